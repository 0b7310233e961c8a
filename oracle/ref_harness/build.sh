#!/bin/bash
# oracle/ref_harness/build.sh -- build the reference harness and regenerate the reference-pinned goldens.
# Needs cargo and the crates of Cargo.toml (network or a vendored registry): neither exists in the build image of this
# repository, which is why tests/test_reference_pin.py skips there.  Outputs go to oracle/_ref/ (git-ignored) and, for the
# goldens, tests/golden/reference_*.tbl (commit those: they are data).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
command -v cargo > /dev/null || { echo "cargo not found: this step needs a Rust toolchain" >&2; exit 3; }
mkdir -p "$ROOT/oracle/_ref"
(cd "$HERE" && cargo build --release --target-dir "$ROOT/oracle/_ref/target")
cp "$ROOT/oracle/_ref/target/release/g1s-ref-diff" "$ROOT/oracle/_ref/g1s-ref-diff"
# the fixtures: small seeded Y4M pairs written by the repository's own generator (no reference data involved)
python "$ROOT/tools/make_y4m_fixture.py" "$ROOT/oracle/_ref/fixtures"
for src in "$ROOT"/oracle/_ref/fixtures/*_source.y4m; do
  base=$(basename "$src" _source.y4m)
  "$ROOT/oracle/_ref/g1s-ref-diff" "$src" "${src%_source.y4m}_denoised.y4m" "$ROOT/tests/golden/reference_${base}.tbl"
  echo "wrote tests/golden/reference_${base}.tbl"
done
