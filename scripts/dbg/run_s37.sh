cd /root/repo
timeout 120 python scripts/mb_overlap.py --skip-old --depth 4 --variants 2,514,2,514 --out gpurun_out/s37_mb.json > gpurun_out/s37_mb.log 2>&1
timeout 60 python scripts/mb_overlap.py --skip-old --depth 1 --variants 530 --out gpurun_out/s37t_mb.json > gpurun_out/s37t_mb.log 2>&1
