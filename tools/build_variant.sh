#!/bin/bash
# A/B builds of the development library (BLS12-381 G1 + Fr only; other curves return ARK_HIP_ERR_ARG):
#   tools/build_variant.sh NAME [extra hipcc flags, e.g. -DARK_LAZY_MIN_WAVES=1]
# -> algebra_amd/variants/libark_hip_NAME.so; select it with ARK_HIP_LIB=<path> (algebra_amd/_lib.py)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/algebra_amd/csrc
O=/tmp/ark_variant_$name
mkdir -p $O $R/algebra_amd/variants
FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -I$S $*"
for u in runtime msm fft comm; do /opt/rocm/bin/hipcc $FL -DARK_HIP_DEV -c $S/capi_$u.hip -o $O/capi_$u.o & done
/opt/rocm/bin/hipcc $FL -c $S/msm_bls12_381_g1.hip -o $O/msm.o &
/opt/rocm/bin/hipcc $FL -c $S/fft_bls12_381_fr.hip -o $O/fft.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/algebra_amd/variants/libark_hip_$name.so $O/capi_runtime.o $O/capi_msm.o $O/capi_fft.o $O/capi_comm.o $O/msm.o $O/fft.o
echo built $R/algebra_amd/variants/libark_hip_$name.so
