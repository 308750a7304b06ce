#!/bin/bash
# One command for a machine that has cargo, hipcc and an MI355X: builds libark_hip.so, vendors the three crates + the test
# crate into a scratch copy of arkworks-rs/algebra 0.6.0, applies patches/0001-0005 in order and runs
#   * the reference's OWN test-suites over the GPU path: `cargo test --features hip` in ark-bls12-381 / ark-bls12-377 /
#     ark-bn254 (test_group!(..; sw) incl. the MSM templates of test-templates/src/msm.rs:8-157) and in ark-poly (the
#     radix-2 domain tests of poly/src/domain/radix2/mod.rs:351-600 through the hook of patches/0003);
#   * rust/ark-hip-tests: the same templates over the wrapper configs of unmodified arkworks and the domain newtype.
#
#     rust/ci.sh <path to an arkworks-rs/algebra checkout at v0.6.0> [scratch dir]
#
# The build image of this repository has no Rust toolchain, so this script has never run there;
# tests/test_rust_boundary.py checks statically what it can: the patches apply in order, the crates it vendors exist where
# their Cargo.toml paths point, every extern "C" declaration matches include/ark_hip.h.
set -euo pipefail
ALGEBRA=${1:?usage: rust/ci.sh <arkworks-rs/algebra checkout> [scratch dir]}
REPO=$(cd "$(dirname "$0")/.." && pwd)
WORK=${2:-$(mktemp -d)}
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain (>= 1.77)"; exit 2; }
[ -f "$REPO/algebra_amd/libark_hip.so" ] || make -C "$REPO/algebra_amd/csrc" -j"$(nproc)"
rm -rf "$WORK/algebra"; mkdir -p "$WORK"; cp -r "$ALGEBRA" "$WORK/algebra"; cd "$WORK/algebra"
cp -r "$REPO/rust/ark-hip-sys" hip-sys
cp -r "$REPO/rust/ark-hip" hip
cp -r "$REPO/rust/ark-hip-curves" curves/hip-configs
cp -r "$REPO/rust/ark-hip-tests" hip-tests
for p in "$REPO"/patches/000*.patch; do git apply -p1 "$p" || patch -p1 < "$p"; done
export ARK_HIP_LIB_DIR="$REPO/algebra_amd" LD_LIBRARY_PATH="$REPO/algebra_amd:${LD_LIBRARY_PATH:-}"
(cd curves && cargo test --release -p ark-bls12-381 -p ark-bls12-377 -p ark-bn254 --features hip)
cargo test --release -p ark-poly --features hip
(cd hip-tests && cargo test --release)
echo "ark-hip: the reference's test-suites passed over the GPU path"
