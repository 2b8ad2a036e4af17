set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py tests/test_recconv_gpu.py -q -x 2>&1 | tail -3
python3 /dev/stdin <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import recnext_amd
dev = torch.device("cuda:0")
for fused in ("1", "0"):
    os.environ["RCX_TRAIN_FUSED"] = fused
    for (n, c, h, level) in [(128, 64, 56, 4), (128, 128, 28, 3)]:
        mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).train()
        x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for _ in range(5): mod(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): mod(x)
        e.record(); torch.cuda.synchronize()
        print("fused" if fused == "1" else "per-step", (n, c, h, level), "training forward", round(s.elapsed_time(e) / 30 * 1e3, 1), "us")
PY
for a in "56 64 256 1" "28 128 256 1"; do timeout -k 5 60 ./tools/cpt_bench $a; done
timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
