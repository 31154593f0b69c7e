"""Assembly edits of k_policy_rollout<8, f16x3, 4> in the failing build: policy_fault_asm.py <edit> <in.s> <out.s>.
The target is the last instruction of the FIRST contact chain: v_pk_mul_f32 vD, vS0, vD op_sel:[0,1] (the only one of that form)."""
import re
import sys

edit, src, dst = sys.argv[1:4]
lines = open(src).read().split("\n")
name = "_ZN3cm316k_policy_rolloutILi8ELi2ELi4EEEvNS_12PolicyParamsE"
a = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
hits = [i for i in range(a, b) if re.search(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\1:\2\] op_sel:\[0,1\]", lines[i])]
assert len(hits) == 1, hits
i = hits[0]
m = re.search(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\]", lines[i])
d0, d1, s0, s1 = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
print("target line %d: %s   (previous: %s | next: %s)" % (i - a, lines[i].strip(), lines[i - 1].strip(), lines[i + 1].strip() or lines[i + 2].strip()))
if edit == "ctrl":
    pass
elif edit == "nop_before":
    lines.insert(i, "\ts_nop 3")
elif edit == "nop_after":
    lines.insert(i + 1, "\ts_nop 3")
elif edit == "two_mul":          # low first (it needs the old high word of src1 = dst), then high
    lines[i:i + 1] = ["\tv_mul_f32_e32 v%d, v%d, v%d" % (d0, s0, d1), "\tv_mul_f32_e32 v%d, v%d, v%d" % (d1, s1, d1)]
elif edit == "pen_copy":         # the same packed multiply with the penetration term copied to a free pair first: no op_sel, no overlap
    lines[i:i + 1] = ["\tv_mov_b32_e32 v88, v%d" % d1, "\tv_mov_b32_e32 v89, v%d" % d1, "\ts_nop 0",
                      "\tv_pk_mul_f32 v[%d:%d], v[%d:%d], v[88:89]" % (d0, d1, s0, s1)]
elif edit == "other_dst":        # op_sel form kept, destination not overlapping src1
    lines[i:i + 1] = ["\tv_pk_mul_f32 v[88:89], v[%d:%d], v[%d:%d] op_sel:[0,1]" % (s0, s1, d0, d1), "\ts_nop 0",
                      "\tv_mov_b32_e32 v%d, v88" % d0, "\tv_mov_b32_e32 v%d, v89" % d1]
elif edit == "fixup_nop":        # wait states between the v_div_fixup that writes src0's low word and the packed multiply's other producers
    lines.insert(i - 1, "\ts_nop 3")
else:
    raise SystemExit("unknown edit " + edit)
open(dst, "w").write("\n".join(lines))
