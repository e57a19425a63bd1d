cd /root/repo
timeout 900 python scripts/mb_overlap.py --skip-old --depth 4 --variants 2 --ablate 0,1,2,4,3,5,6 --out gpurun_out/s34_mb.json > gpurun_out/s34_mb.log 2>&1
tail -3 gpurun_out/s34_mb.log | cut -c1-400
