"""Basic-block instruction census of one kernel in a hipcc `-S --cuda-device-only` listing (tools: compute rooflines of the
VALU-bound kernels, e.g. moq_mse_sweep -- VERDICT round 4, next #7).

  hipcc --offload-arch=gfx950 -O3 <flags of csrc/build.sh> -S --cuda-device-only moq_calib.hip -o /tmp/moq_calib.s
  python tools/isa_blocks.py /tmp/moq_calib.s '_ZN3moq16mse_group_kernelILi2ELi16ELb0EE' [.LBBn_m]

Prints, per basic block: VALU / transcendental / SALU / VMEM / LDS / other instruction counts and the block's branch targets,
so the loop bodies (blocks that branch back to themselves or to an earlier label) can be read off; with a label, that
block's opcode histogram."""
import collections
import re
import sys


def census(path, prefix):
    blocks, cur, inside = [], None, False
    for line in open(path):
        if not inside:
            if line.startswith(prefix) and ":" in line:
                inside = True
                cur = {"label": "entry", "n": collections.Counter(), "ops": collections.Counter(), "to": []}
                blocks.append(cur)
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = {"label": m.group(1), "n": collections.Counter(), "ops": collections.Counter(), "to": []}
            blocks.append(cur)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        cur["ops"][op] += 1
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", op):
            cur["n"]["trans"] += 1
        elif op.startswith("v_"):
            cur["n"]["valu"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["to"].append(s.split()[-1])
            cur["n"]["salu"] += 1
        elif op.startswith("s_"):
            cur["n"]["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["n"]["vmem"] += 1
        elif op.startswith("ds_"):
            cur["n"]["lds"] += 1
        else:
            cur["n"]["other"] += 1
    return blocks


if __name__ == "__main__":
    bl = census(sys.argv[1], sys.argv[2])
    labels = [b["label"] for b in bl]
    for i, b in enumerate(bl):
        back = [t for t in b["to"] if t in labels and labels.index(t) <= i]
        print(f"{b['label']:12s} valu {b['n']['valu']:4d} trans {b['n']['trans']:3d} salu {b['n']['salu']:3d} vmem {b['n']['vmem']:3d} "
              f"lds {b['n']['lds']:3d} other {b['n']['other']:3d}  -> {','.join(b['to'])}{'   <== LOOP' if back else ''}")
    if len(sys.argv) > 3:
        for b in bl:
            if b["label"] == sys.argv[3]:
                for op, n in sorted(b["ops"].items(), key=lambda kv: -kv[1]):
                    print(f"    {op:28s} {n}")
