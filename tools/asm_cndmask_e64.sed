# v_cndmask_b32 in its 32-bit encoding reads VCC implicitly; runs of them issue at ~19 SIMD-cycles each on gfx950 once four waves
# share a CU (profiles/r05/micro/issue_rates.txt), the 64-bit encoding of the very same operation (VCC named as the SGPR pair) at 4
s/^(\s*)v_cndmask_b32_e32 (v[0-9]+, [^,]+, v[0-9]+), vcc$/\1v_cndmask_b32_e64 \2, vcc/
