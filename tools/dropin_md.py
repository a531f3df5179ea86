"""tools/dropin_bench.py's JSON -> the markdown table committed under profiles/ (rows: format x mode)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"# What the drop-in buys: wall seconds of the reference's own `mtq.quantize` / `export_hf_checkpoint`\n")
print(f"`tools/dropin_bench.py`: {d['what']}; reference = {d['reference']} archive of the checkout, run unmodified.\n")
print("| format | mode | quantize s | vs eager | seam calls inside quantize() | export s | result vs the first row |")
print("|---|---|---|---|---|---|---|")
names = {"eager": "reference eager (ROCm: no extension)", "kernels": "+ `install()` (kernel seams S1 / S3 / S6)",
         "algorithms": "+ `install(algorithms=True)` (S7)", "mirror": "this package's own `quantize()`"}
for fmt, rows in d["formats"].items():
    base = next((r.get("quantize_s") for r in rows.values() if "quantize_s" in r), None)
    for mode, r in rows.items():
        if "quantize_s" not in r:
            print(f"| {fmt} | {names[mode]} | -- | | | | {r.get('skipped') or r.get('error')} |")
            continue
        cmp_ = next((v for k, v in r.items() if k.startswith("vs_")), None)
        how = "(first row)"
        if cmp_:
            how = f"{cmp_['identical']} / {cmp_['tensors']} quantizer tensors identical"
            if cmp_.get("alpha_picks"):
                how += f", {cmp_['alpha_picks_equal']} / {cmp_['alpha_picks']} alpha picks equal"
            if cmp_["identical"] != cmp_["tensors"]:
                how += f" (others within {cmp_['worst_relative_difference']:.1e} relative)"
        calls = r.get("seam_calls_in_quantize", "")
        s7 = ", ".join(f"{k} x{v}" for k, v in (r.get("s7") or {}).items())
        print(f"| {fmt} | {names[mode]} | {r['quantize_s']:.3f} | {base / r['quantize_s']:.2f}x | {calls}{' (' + s7 + ')' if s7 else ''} | "
              f"{r.get('export_s', '')} | {how} |")
