#!/bin/bash
# Runs the pinning test against a checkout of NP-Eng/ligero and leaves tests/golden/rust_dump.json in this repository.
#   LIGERO_REFERENCE=/path/to/ligero ./rust-shim/run.sh
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
ref=${LIGERO_REFERENCE:?set LIGERO_REFERENCE to a checkout of NP-Eng/ligero}
ref=$(cd "$ref" && pwd)
work=$(mktemp -d)
cp -r "$here"/. "$work"/
sed -i "s|path = \"../../reference\"|path = \"$ref\"|" "$work/Cargo.toml"
# the reference reads its fixtures relative to the working directory (src/ligero/tests.rs:367-378): run from its root
( cd "$ref" && LIGERO_PIN_OUT="$here/../tests/golden/rust_dump.json" cargo test --manifest-path "$work/Cargo.toml" --release -- --nocapture --test-threads 1 )
echo "wrote $here/../tests/golden/rust_dump.json; now: python -m pytest tests/test_rust_pin.py -q"
