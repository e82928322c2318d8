#!/usr/bin/env python3
"""Per-kernel ISA summary of a `hipcc -S --cuda-device-only` listing: VGPRs, instruction counts by class, memory
instructions and non-temporal hints.  usage: isa_stats.py listing.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
RX = {"valu": r"^\s+v_", "salu": r"^\s+s_", "mfma": r"^\s+v_mfma", "ds": r"^\s+ds_", "gload": r"^\s+(global|buffer)_load",
      "gstore": r"^\s+(global|buffer)_store", "nt": r" nt", "waitcnt": r"s_waitcnt", "scratch": r"scratch_"}
for m in re.finditer(r"^(\w+):\s*; @\1\n", s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    j = s.index(".end_amdhsa_kernel", m.end())
    body = s[m.end():j]
    code = body[:body.index("s_endpgm")] if "s_endpgm" in body else body
    meta = {k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [None, "?"])[1]
            for k in ("next_free_vgpr", "accum_offset", "group_segment_fixed_size")}
    counts = " ".join(f"{k} {len(re.findall(rx, code, re.M))}" for k, rx in RX.items())
    print(f"{name[:110]}\n   vgpr {meta['next_free_vgpr']} accum_off {meta['accum_offset']} lds {meta['group_segment_fixed_size']}  {counts}")
