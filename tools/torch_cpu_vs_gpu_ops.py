"""Where torch's own small-vector ops differ between the host and the MI355X (the reason numerics.scale_math exists, and why
the reference writes a different checkpoint from a GPU run than from a CPU run): tensor / python scalar, tensor / 0-dim
tensor in 16-bit, the FP8 cast of a 16-bit quotient, reciprocal, pow.  Prints one JSON line per op: elements compared,
elements whose bits differ, largest difference in ulps, and the first differing example."""
import json

import torch


def ulps(a, b):
    ia = a.float().contiguous().view(torch.int32).long()
    ib = b.float().contiguous().view(torch.int32).long()
    return (ia - ib).abs()


def report(name, cpu, gpu, inputs=None):
    gpu = gpu.cpu()
    same_bits = cpu.contiguous().view(torch.uint8).reshape(cpu.shape + (-1,)) == gpu.contiguous().view(torch.uint8).reshape(cpu.shape + (-1,))
    bad = ~same_bits.all(-1)
    line = {"op": name, "n": cpu.numel(), "differ": int(bad.sum())}
    if bad.any():
        i = int(bad.reshape(-1).nonzero()[0])
        if cpu.dtype in (torch.float32, torch.bfloat16, torch.float16):
            line["max_ulp"] = int(ulps(cpu, gpu).max()) if cpu.dtype == torch.float32 else None
        line["first"] = {"index": i, "cpu": float(cpu.reshape(-1)[i].float()), "gpu": float(gpu.reshape(-1)[i].float())}
        if inputs is not None:
            line["first"]["inputs"] = [float(t.reshape(-1)[i % t.numel()].float()) for t in inputs]
    print(json.dumps(line), flush=True)


def main():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1 << 20, generator=g) * 4 + 1e-3
    dev = "cuda"
    report("f32 tensor / 448.0 (python scalar)", x / 448.0, x.to(dev) / 448.0, [x])
    report("f32 tensor / 7.0", x / 7.0, x.to(dev) / 7.0, [x])
    report("f32 tensor / int 64", x / 64, x.to(dev) / 64, [x])
    d = torch.rand(1 << 20, generator=g) + 0.5
    report("f32 tensor / f32 tensor", x / d, x.to(dev) / d.to(dev), [x, d])
    report("1.0 / f32 tensor", 1.0 / x, 1.0 / x.to(dev), [x])
    report("f32 reciprocal()", x.reciprocal(), x.to(dev).reciprocal(), [x])
    report("f32 pow(0.3)", x.pow(0.3), x.to(dev).pow(0.3), [x])
    report("f32 pow(0.5)", x.pow(0.5), x.to(dev).pow(0.5), [x])
    report("f32 sqrt", x.sqrt(), x.to(dev).sqrt(), [x])
    w = (torch.randn(1 << 20, generator=g) * 0.02).to(torch.bfloat16)
    s0 = torch.tensor(float(w.float().abs().max()) / 448.0)  # 0-dim fp32, as torch.tensor(python float)
    report("bf16 tensor / 0-dim f32 tensor", w / s0, w.to(dev) / s0.to(dev), [w])
    q_cpu, q_gpu = w / s0, w.to(dev) / s0.to(dev)
    report("(bf16 / 0-dim).to(float8_e4m3fn)", q_cpu.to(torch.float8_e4m3fn), q_gpu.to(torch.float8_e4m3fn), [w, q_cpu])
    report("same bf16 quotient .to(float8_e4m3fn)", q_cpu.to(torch.float8_e4m3fn), q_cpu.to(dev).to(torch.float8_e4m3fn), [q_cpu])
    xf = torch.randn(1 << 20, generator=g) * 200
    report("f32 .to(float8_e4m3fn)", xf.clamp(-448, 448).to(torch.float8_e4m3fn), xf.to(dev).clamp(-448, 448).to(torch.float8_e4m3fn), [xf])
    h = (torch.randn(1 << 20, generator=g) * 0.02).half()
    report("f16 tensor / 0-dim f32 tensor", h / s0, h.to(dev) / s0.to(dev), [h])
    report("f16 quotient .to(float8_e4m3fn)", (h / s0).to(torch.float8_e4m3fn), (h / s0).to(dev).to(torch.float8_e4m3fn), [h])
    a0 = torch.tensor(0.37, dtype=torch.bfloat16)
    m = torch.tensor(1.7)
    report("bf16 0-dim amax * f32 0-dim multiplier", (a0 * m).reshape(1), (a0.to(dev) * m.to(dev)).reshape(1))
    wf = w.float()
    sc = torch.rand(1 << 20, generator=g) * 1e-3 + 1e-4
    report("(f32 / f32).round()", (wf / sc).round(), (wf.to(dev) / sc.to(dev)).round(), [wf, sc])
    report("bf16 * f32 vector -> bf16", (w * sc[: w.numel()]).to(torch.bfloat16), (w.to(dev) * sc.to(dev)).to(torch.bfloat16), [w, sc])


if __name__ == "__main__":
    main()
