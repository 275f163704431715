#!/bin/bash
# MI355X issue-port / latency microbenchmarks behind DESIGN.md's cost model (dev tools).
cd "$(dirname "$0")"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w "$f" -o "${f%.hip}" || exit 1; done
