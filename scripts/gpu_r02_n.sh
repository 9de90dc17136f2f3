#!/bin/bash
export TMPDIR=/tmp
scripts/ab_libs.sh C3 10 p2 p1 p2w7 p2w5
scripts/ab_libs.sh C2 20 base p2
