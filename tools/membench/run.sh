# usage (on the GPU box): bash tools/membench/run.sh [frames]
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/membench membench.hip
timeout -k 10 120 /tmp/membench ${1:-128}
