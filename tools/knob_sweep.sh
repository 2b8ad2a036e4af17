#!/bin/bash
# rocprofv3 kernel time of one block shape under the launch knobs (RCX_LANES_NI x RCX_LANES_WAVES).  usage: tools/knob_sweep.sh N C H LEVEL
N=$1; C=$2; H=$3; L=$4
for ni in 1 2; do for w in 4 8; do
  export RCX_LANES_NI=$ni RCX_LANES_WAVES=$w
  echo -n "ni=$ni waves=$w  "; tools/sweep_batch.sh $C $H $L "$N" | cut -c1-40
done; done
