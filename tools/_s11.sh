mkdir -p gpurun_out/s11
for w in 1 2 4 8; do echo "== world $w"; timeout -k 10 200 python tools/ab_option.py 11 $w 0,2 2>&1 | tail -2; done > gpurun_out/s11/ab11.txt 2>&1
cat gpurun_out/s11/ab11.txt
for w in 2 4 8; do echo "== world $w (option 7: 0 off, 1 always fused solve+update)"; timeout -k 10 200 python tools/ab_option.py 7 $w 0,1 2>&1 | tail -2; done > gpurun_out/s11/ab7.txt 2>&1
cat gpurun_out/s11/ab7.txt
