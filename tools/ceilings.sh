#!/bin/bash
# the write ceilings of the evaluation's launch shapes (tools/ceiling_probe.hip) -> profiles/r04_write_ceiling.txt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ceiling_probe tools/ceiling_probe.hip || exit 1
out=${1:-gpurun_out/r4_write_ceiling.txt}
{
echo "# tools/ceiling_probe <bytes> <workgroups> <threads>: hand-written store-only and read-10%/write-90% kernels with the"
echo "# evaluation's own grid; mean of 400 back-to-back launches between two HIP events"
echo "## config 2: hypersensitive 10 001 nodes, 1.6 MB, 167 tiles x 4 waves"
tools/ceiling_probe 1600080 167 256
echo "## config 3: cart-pole 15 001 nodes, 6.8 MB, 239 tiles x 4 waves"
tools/ceiling_probe 6840000 239 256
echo "## config 4: shuttle 60 001 nodes, 82 MB, 953 tiles x 2 waves"
tools/ceiling_probe 82080320 953 128
echo "## config 5 (uniform): Delta III 4 x 12 501 nodes, 85.6 MB, 896 tiles x 2 waves"
tools/ceiling_probe 85602400 896 128
echo "## config 5 (ph-refined mesh, ~50 k nodes): 100 MB, 900 tiles x 2 waves / x 1 wave"
tools/ceiling_probe 100000000 900 128
tools/ceiling_probe 100000000 900 64
echo "## hypersensitive 1 000 001 nodes, 160 MB, 3922 tiles x 256 threads"
tools/ceiling_probe 160000080 3922 256
echo "## shuttle 600 001 nodes, 821 MB, 2353 tiles x 256 threads"
tools/ceiling_probe 820800000 2353 256
echo "## the same bytes with the chip filled (2048 workgroups x 256 threads)"
tools/ceiling_probe 82080320 2048 256
tools/ceiling_probe 160000080 2048 256
tools/ceiling_probe 820800000 2048 256
} > $out 2>&1
cat $out
