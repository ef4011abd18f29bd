export TMPDIR=/tmp
# (round 5: the RAYHS_* knobs this recipe sets exist only in -DRAYHS_DEV_KNOBS builds: build its libraries with tools/build_exp.sh <name> [flags], and use build_exp/cur -- tools/build_exp.sh cur -- where it says "default lib")
for leaf in 2 3 4 5; do for wl in dragon dragon_low; do echo "== leaf $leaf $wl"; RAYHS_BVH_LEAF=$leaf timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | tail -1 | sed 's/alg_GB.*nodes/nodes/'; done; done
