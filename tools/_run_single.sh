cd /root/repo
timeout -k 10 300 python3 tools/split_check.py 256x384x256x10 1024x4096x256x10 4096x350x200x10 300x500x129x5
echo "--- gold batched: double vs single image"
for b in 1 16; do NMF_SPLIT_DOUBLE=1 python3 tools/small_iter.py 4096 350 128 1 128 1 0 0 $b 2>&1 | tail -1; python3 tools/small_iter.py 4096 350 128 1 128 1 0 0 $b 2>&1 | tail -1; done
NMF_SPLIT_SINGLE=1 python3 tools/small_iter.py 4096 350 128 1 128 1 0 0 1 2>&1 | tail -1
timeout -k 10 400 python3 tools/crossover.py 1024x4096x256 4096x350x256 4096x1024x256 2048x4096x256 4096x4096x256 1024x1024x200 512x512x256
