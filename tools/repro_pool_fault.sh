#!/bin/bash
# the four guard combinations of the pooled BatchNorm-backward apply pass beside one LDS-DMA weight-gradient launch (tools/repro_pool_fault.py)
cd $GRAFT_REPO_ROOT
N=${1:-200}
for d in 0 1; do for a in 0 1; do RPNET_BN_POOL_DRAIN=$d RPNET_BN_POOL_ALONE=$a timeout 300 python tools/repro_pool_fault.py $N 2>&1 | grep -v amdgpu.ids | tail -1; done; done
