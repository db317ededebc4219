#!/bin/bash
# builds and runs the matrix-pipe probe; rocm-smi is sampled while it runs (clock and package power under load)
here=$(cd "$(dirname "$0")" && pwd)
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak "$here/mfma_peak.hip" || exit 1
( for i in 1 2 3 4 5 6; do sleep 0.4; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power\|Socket Power" | head -2 | tr '\n' ' '; echo; done ) > /tmp/mfma_peak_smi.txt &
/tmp/mfma_peak "${1:-40}"
wait
echo "rocm-smi while the probe ran:"; cat /tmp/mfma_peak_smi.txt
