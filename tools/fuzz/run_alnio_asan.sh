#!/bin/bash
# CPU-only: the alignment reader under AddressSanitizer + UBSan on valid, truncated and bit-flipped
# BAM / SAM inputs (made from the reference's test data set).  Usage: tools/fuzz/run_alnio_asan.sh
set -e
cd "$(dirname "$0")/../.."
W=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Imiso_amd/csrc \
    tools/fuzz/alnio_driver.cpp miso_amd/csrc/alnio.cpp -o $W/drv -lz -lpthread
python3 - "$W" <<'PY'
import sys, gzip, random
sys.path.insert(0, "tests")
from _bam import sam_to_bam
W = sys.argv[1]
lines = gzip.open("tests/golden/data/c2c12.Atp2b1.sam.gz", "rt").read().splitlines()
small = "\n".join(lines[:460]) + "\n"
open(W + "/ok.sam", "w").write(small)
sam_to_bam(small, W + "/ok.bam", block=5000)
raw = open(W + "/ok.bam", "rb").read()
rng = random.Random(1)
for n, cut in enumerate(range(1, len(raw), max(1, len(raw) // 150))):
    open(W + "/t%03d.bam" % n, "wb").write(raw[:cut])
for k in range(200):
    b = bytearray(raw)
    for _ in range(rng.randint(1, 4)):
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    open(W + "/f%03d.bam" % k, "wb").write(bytes(b))
for k in range(100):
    b = bytearray(small.encode())
    for _ in range(rng.randint(1, 6)):
        b[rng.randrange(len(b))] = rng.randrange(256)
    open(W + "/s%03d.sam" % k, "wb").write(bytes(b))
PY
(cd $W && ./drv ok.sam ok.bam t*.bam f*.bam s*.sam)
rm -rf $W
