set -e
cd $GRAFT_REPO_ROOT
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I pl-viwo_amd/csrc -I include"
hipcc $F tools/ubench/bchol_time.hip -o /tmp/bt_new.bin
hipcc $F -DPLV_BC_FOLLOW_BARRIER tools/ubench/bchol_time.hip -o /tmp/bt_old.bin
hipcc $F -DNO_STAMPS tools/ubench/bchol_time.hip -o /tmp/bt_new_ns.bin
hipcc $F -DNO_STAMPS -DPLV_BC_FOLLOW_BARRIER tools/ubench/bchol_time.hip -o /tmp/bt_old_ns.bin
echo "=== default (strips follow the chain live), stamps"; /tmp/bt_new.bin
echo "=== -DPLV_BC_FOLLOW_BARRIER (chain first, barrier, strip_follow), stamps"; /tmp/bt_old.bin
echo "=== default, no stamps"; /tmp/bt_new_ns.bin | grep -E "us|finger"
echo "=== -DPLV_BC_FOLLOW_BARRIER, no stamps"; /tmp/bt_old_ns.bin | grep -E "us|finger"
