#!/bin/bash
# Register / spill / scratch figures of the trace kernels from the code-object metadata (development aid):
#   tools/kinfo.sh [extra hipcc flags, e.g. -DRAYHS_WAVES_PER_EU=3]
set -e
d=$(mktemp -d); cd "$d"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DRAYHS_KERNEL_SUFFIX=exact "$@" --cuda-device-only -save-temps \
  -c /root/repo/rayhs_amd/csrc/trace_kernel.hip -I/root/repo/rayhs_amd/csrc -o trace.co 2>&1 | grep -v warning | head -5
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" *.s | sed 's/^[^:]*://' | paste - - - - - - | sed 's/ \+/ /g' | grep trace
awk '/^rayhs_trace_exact:/{p=1} /^rayhs_trace_wide_exact:/{p=0} p' *.s > /tmp/prod.s
echo "prod.s: $(wc -l < /tmp/prod.s) lines, v_readlane $(grep -c v_readlane /tmp/prod.s), v_writelane $(grep -c v_writelane /tmp/prod.s), scratch $(grep -c scratch_ /tmp/prod.s)"
