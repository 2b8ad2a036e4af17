mkdir -p gpurun_out/h40
timeout -k 10 60 tools/cplbwd_probe > gpurun_out/h40/out.txt 2>&1
echo rc=$? >> gpurun_out/h40/out.txt
