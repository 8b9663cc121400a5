#!/bin/bash
# the multi-unknown stencils of profiles/r03_stencils.txt (template blocks whose rows do not hold every diagonal)
set -e -o pipefail
for a in "128 128 128 7 3" "100 100 100 7 4" "140 140 140 7 2" "1500 1500 1 27 3" "200 200 200 27 1" "128 128 488 27 1"; do
    python3 scripts/stencil27.py $a | grep -v "^CG"
done
