#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/profile_shapes.sh r06_k ref160 c2 c1 2>&1 | tail -30
