#!/bin/bash
# round 3, session 3, call 18: the other BASELINE configs on the round's final code (the vector-column dW only for trunks >= 512 wide)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_preset.sh r3d_blender_256 --preset blender_256
bash tools/profile_preset.sh r3d_llff_raw --preset llff_raw
bash tools/profile_preset.sh r3d_blender_refnerf --preset blender_refnerf
