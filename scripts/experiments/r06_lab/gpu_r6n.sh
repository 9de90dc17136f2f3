#!/bin/bash
# r6n: phase clocks of the de-spilled count kernel, C4 and C5 (20 000 spectra)
export TMPDIR=/tmp
timeout 900 python scripts/tile_phase_cfg.py C4 20000 2>&1 | tail -11
timeout 900 python scripts/tile_phase_cfg.py C5 20000 2>&1 | tail -11
