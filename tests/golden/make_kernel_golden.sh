#!/bin/bash
# Regenerates tests/golden/kernel_functions.npz from the REFERENCE's own kernel functions (build container only: needs /root/reference).
# 1. The reference's generated configuration headers are produced by its own templates (src/ChConfig.h.in, src/ChVersion.h.in,
#    src/chrono_fsi/sph/ChFsiConfigSPH.h.in) through cmake's configure_file with no option set -- single precision off ... i.e. Real = float,
#    nothing optional enabled: the configuration the survey's probe used -- into a scratch directory outside the repository.
# 2. kernel_functions_gen.cpp includes src/chrono_fsi/sph/physics/SphGeneral.cuh in place and is compiled as HOST code by hipcc with
#    -ffp-contract=off (the parity contract's arithmetic: products and sums rounded separately), then run.
# 3. The float stream is packed into the .npz (inputs and outputs: data, no source text).
set -e
here=$(cd "$(dirname "$0")" && pwd)
ref=${REFERENCE_ROOT:-/root/reference}
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/inc/chrono" "$tmp/inc/chrono_fsi/sph"
cat > "$tmp/configure.cmake" <<CM
configure_file("$ref/src/ChConfig.h.in" "$tmp/inc/chrono/ChConfig.h" @ONLY)
configure_file("$ref/src/ChVersion.h.in" "$tmp/inc/chrono/ChVersion.h" @ONLY)
configure_file("$ref/src/chrono_fsi/sph/ChFsiConfigSPH.h.in" "$tmp/inc/chrono_fsi/sph/ChFsiConfigSPH.h" @ONLY)
CM
cmake -P "$tmp/configure.cmake"
/opt/rocm/bin/hipcc -x hip -std=c++17 --offload-arch=gfx950 -O2 -ffp-contract=off -DCHRONO_USE_HIP -I"$tmp/inc" -I"$ref/src" -I/opt/rocm/include \
    "$here/kernel_functions_gen.cpp" -o "$tmp/gen"
"$tmp/gen" > "$tmp/out.f32"
python3 - "$tmp/out.f32" "$here/kernel_functions.npz" <<'PY'
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.float32).reshape(4, 3, 96, 2 + 3 * 6)
np.savez_compressed(sys.argv[2], h=np.array([0.01, 0.012, 0.025], dtype=np.float32), r=a[..., 0], W=a[..., 1],
                    d=a[..., 2:].reshape(4, 3, 96, 3, 6)[..., :3], gradW=a[..., 2:].reshape(4, 3, 96, 3, 6)[..., 3:])
print("kernel_functions.npz:", {k: v.shape for k, v in np.load(sys.argv[2]).items()})
PY
