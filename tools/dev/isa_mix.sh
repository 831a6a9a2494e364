#!/bin/bash
# dev: static instruction mix of one kernel of tsf_inst_quad3.hip (default: the 12-wave register-weights kernel)
# usage: tools/dev/isa_mix.sh [mangled-name-fragment] [extra hipcc flags...]
K=${1:-ILi28ELi1ELi12ELi0ELi56ELb0ELb1ELi12ELb0E}; shift
cd "$(dirname "$0")/../../time_series_spark_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -mllvm -disable-machine-licm "$@" --cuda-device-only -S tsf_inst_quad3.hip -o /tmp/isa_mix.s 2>/dev/null
S=$(grep -n "^_ZN3tsf15fit_quad_kernel$K.*:" /tmp/isa_mix.s | head -1 | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /s_endpgm/ {print NR; exit}' /tmp/isa_mix.s)
awk -v s=$S -v e=$E 'NR>s && NR<e' /tmp/isa_mix.s > /tmp/isa_mix_k.s
printf "valu %s salu %s lds %s vmem %s | readlane %s writelane %s nop %s v_mov %s dpp %s fp64 %s cndmask %s cmp %s branch %s waitcnt %s\n" \
  $(grep -c "^\s*v_" /tmp/isa_mix_k.s) $(grep -c "^\s*s_" /tmp/isa_mix_k.s) $(grep -c "^\s*ds_" /tmp/isa_mix_k.s) $(grep -c "^\s*global_\|^\s*scratch_" /tmp/isa_mix_k.s) \
  $(grep -c "v_readlane" /tmp/isa_mix_k.s) $(grep -c "v_writelane" /tmp/isa_mix_k.s) $(grep -c "s_nop" /tmp/isa_mix_k.s) $(grep -c "v_mov_b32_e32\|v_mov_b64" /tmp/isa_mix_k.s) $(grep -c "_dpp" /tmp/isa_mix_k.s) \
  $(grep -c "v_fma_f64\|v_fmac_f64\|v_add_f64\|v_mul_f64" /tmp/isa_mix_k.s) $(grep -c "v_cndmask" /tmp/isa_mix_k.s) $(grep -c "v_cmp" /tmp/isa_mix_k.s) $(grep -c "s_cbranch" /tmp/isa_mix_k.s) $(grep -c "s_waitcnt" /tmp/isa_mix_k.s)
