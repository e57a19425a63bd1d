cd /root/repo
timeout 900 python scripts/mb_overlap.py --skip-old --depth 4 --variants 2 --ablate 0,8,9,12 --out gpurun_out/s35_mb.json > gpurun_out/s35_mb.log 2>&1
timeout 900 python scripts/mb_overlap.py --skip-old --depth 1 --variants 18 --ablate 8,5 --out gpurun_out/s35t_mb.json > gpurun_out/s35t_mb.log 2>&1
