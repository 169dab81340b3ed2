#!/bin/bash
# TEST INFRASTRUCTURE — stages the UNMODIFIED reference page script for the live-reference harness (oracle/live/) so that
# it travels to the GPU box with the gpurun snapshot: /root/reference does not exist there, and bench.py's `cpu_baseline`
# leg times the reference ITSELF (Chromium + SwiftShader from the kaleido package) on the GPU box's host cores.
# Output only into oracle/_ref/ — git-ignored (the reference's sources never enter the history), not gpurun-ignored.
# The files are byte-for-byte copies (sha256 recorded in oracle/_ref/MANIFEST); the MIT licence travels with them.
set -e
SRC=${1:-/root/reference}
DST="$(cd "$(dirname "$0")" && pwd)/_ref"
if [ ! -f "$SRC/script.js" ]; then
  echo "stage_reference: $SRC/script.js not found (not the build container): keeping whatever is in $DST"
  exit 0
fi
mkdir -p "$DST"
for f in script.js dat.gui.min.js LICENSE; do cp "$SRC/$f" "$DST/$f"; done
( cd "$DST" && sha256sum script.js dat.gui.min.js LICENSE > MANIFEST )
echo "staged $(wc -l < "$DST/script.js") lines of script.js into $DST"
