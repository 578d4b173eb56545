#!/bin/sh
# builds tools/epnp_ord_check (the GPU-side parity harness of csrc/svo_epnp_ord_dev.h; the oracle object is the checker)
set -e
cd "$(dirname "$0")"
make -s -C ../oracle
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -c -o /tmp/epnp_ord_check.o epnp_ord_check.hip 2>&1 | grep -E "error|note: " || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o epnp_ord_check /tmp/epnp_ord_check.o ../oracle/orc_pnp_cv.o
