"""Per-kernel instruction / resource summary of a hipcc -S listing:  python tools/asm_stats.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
heads = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
for i, (pos, name) in enumerate(heads):
    end = heads[i + 1][0] if i + 1 < len(heads) else len(s)
    body = s[pos:end]
    if flt and flt not in name:
        continue
    def c(p):
        return len(re.findall(p, body))
    def meta(k):
        m = re.search(r'\.amdhsa_' + k + r' (\d+)', body)
        return m.group(1) if m else "?"
    sc = re.search(r'; ScratchSize: (\d+)', body)
    print(f"{name[:90]}\n   vgpr {meta('next_free_vgpr')} sgpr {meta('next_free_sgpr')} lds {meta('group_segment_fixed_size')} scratch {sc.group(1) if sc else '?'} | "
          f"mfma {c('v_mfma')} exp {c('v_exp_f')} max3 {c('v_max3_f32')} max {c('v_max_f32')} pk_fma {c('v_pk_fma_f32')} pk_mul {c('v_pk_mul_f32')} pk_add {c('v_pk_add_f32')} "
          f"cvt_pk {c('v_cvt_pk')} glds {c('global_load_lds')} gload {c('global_load_dword')} tr {c('ds_read_b64_tr')} b128 {c('ds_read_b128')} dsw {c('ds_write')} perm {c('v_permlane')} "
          f"saveexec {c('s_and_saveexec')} waitcnt {c('s_waitcnt')} barrier {c('s_barrier')} total {len(re.findall(chr(10) + chr(9) + r'[vsdgb]_', body))}")
