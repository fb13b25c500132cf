#!/bin/bash
# same-box A/B of the data-parallel step's layout with ONE rank (GM3D_FORCE_DIST=1): tools/ab_dist.sh OUTDIR "label:switches" ...
export GM3D_FORCE_DIST=1
exec bash $(dirname $0)/ab.sh "$@"
