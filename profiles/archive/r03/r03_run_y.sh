#!/bin/bash
# round 3, GPU call Y: bit-reproducibility of the mesh step at config-5 size
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_mesh_hip.py -m gpu -q -x -k "bit_reproducible" < /dev/null 2>&1 | tail -4
