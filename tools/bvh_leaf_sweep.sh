export TMPDIR=/tmp
for leaf in 2 3 4 5; do for wl in dragon dragon_low; do echo "== leaf $leaf $wl"; RAYHS_BVH_LEAF=$leaf timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | tail -1 | sed 's/alg_GB.*nodes/nodes/'; done; done
