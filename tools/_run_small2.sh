cd /root/repo
timeout -k 10 300 python3 tools/split_check.py 256x384x64x10 1024x4096x64x20 4096x350x128x20 512x3445x30x20 100x77x5x10 333x1000x100x10
python3 tools/split_slope.py 64
for nw in 2 4 8; do python3 tools/small_iter.py 1024 4096 64 1 400 1 1 $nw; done 2>&1 | grep "it/s"
for nh in 8 11 16; do python3 tools/small_iter.py 4096 350 128 1 400 1 $nh 1; done 2>&1 | grep "it/s"
for nw in 7 14; do python3 tools/small_iter.py 512 3445 30 1 400 1 1 $nw; done 2>&1 | grep "it/s"
for b in 4 16; do python3 tools/small_iter.py 1024 4096 64 1 200 1 0 0 $b; done 2>&1 | grep "it/s"
