#!/bin/bash
# Build the reference's own binary and run its own CLI with --wav-steps (see README.md).
# usage: run_reference.sh REFERENCE_DIR REPO_ROOT
set -euo pipefail
REFERENCE="${1:-/root/reference}"
ROOT="${2:-$(cd "$(dirname "$0")/../.." && pwd)}"
REF_OUT="$ROOT/oracle/_ref"
command -v cargo >/dev/null || { echo "oracle/_ref: no cargo on PATH - the reference (Rust) cannot be built here, skipped"; exit 0; }
[ -f "$REFERENCE/Cargo.toml" ] || { echo "oracle/_ref: no reference checkout at $REFERENCE, skipped"; exit 0; }
mkdir -p "$REF_OUT/home" "$REF_OUT/out"
export HOME="$REF_OUT/home" XDG_CONFIG_HOME="$REF_OUT/home/.config"
cargo build --release --locked --no-default-features --manifest-path "$REFERENCE/Cargo.toml" --target-dir "$REF_OUT/target"
BIN="$REF_OUT/target/release/noaa-apt"
python3 "$ROOT/oracle/ref_harness/make_inputs.py"
for wav in "$REF_OUT"/in/*.wav; do
    name="$(basename "$wav" .wav)"
    for sync in sync nosync; do
        d="$REF_OUT/out/${name}_$sync"
        rm -rf "$d"; mkdir -p "$d"
        flag=""; [ "$sync" = nosync ] && flag="--no-sync"
        # -c minmax: no telemetry needed for the image stage (a noise file has none); the step files are
        # written before the contrast stage either way
        (cd "$d" && "$BIN" -q "$wav" -o decoded.png --wav-steps -p standard -c minmax $flag) || echo "reference failed on $name ($sync)" > "$d/FAILED"
    done
done
# the resample tool on the fixture (test/test.sh:50-51)
d="$REF_OUT/out/noise_fixture_resample"; rm -rf "$d"; mkdir -p "$d"
(cd "$d" && "$BIN" -q "$REF_OUT/in/noise_fixture.wav" -r 80000 -o up_80000.wav && "$BIN" -q "$REF_OUT/in/noise_fixture.wav" -r 11025 -o down_11025.wav)
echo "reference dumps in $REF_OUT/out"
