cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 600 2>&1 | tail -15
for nh in 8 11 16; do python3 tools/small_iter.py 4096 350 128 1 400 1 $nh 1; done 2>&1 | grep "it/s"
python3 tools/small_iter.py 1024 4096 64 1 400 1 1 4 2>&1 | grep "it/s"
python3 tools/small_iter.py 512 3445 30 1 400 1 1 7 2>&1 | grep "it/s"
for b in 2 16; do python3 tools/small_iter.py 1024 4096 64 1 200 1 0 0 $b; done 2>&1 | grep "it/s"
python3 tools/split_slope.py 128
