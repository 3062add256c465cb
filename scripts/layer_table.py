"""Per-launch table of a workload's conv kernels from `bench.py --dump-launches` (serial HIP-event times of one instrumented step).
usage: layer_table.py dump.json > table.txt"""
import collections
import json
import sys

d = json.load(open(sys.argv[1]))
agg = collections.OrderedDict()
for rec in d:
    name, ms, args = rec[0], rec[1], rec[2]
    if name in ("mdcv_conv2d", "mdcv_conv2d_dgrad_bnsums"):
        B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW, stride, pad, dil = args[:12]
        mode = args[-1]
        key = ("fwd" if mode == 0 else ("dgrad+bnsums" if name.endswith("bnsums") else "dgrad"), B, Hin, Cin, Hout, Nout, KH, stride, dil)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
    elif name == "mdcv_pw_bwd":          # 1x1 data gradient + weight-gradient slabs in one launch: (M, Cin, Cout, slabs, addsrc, fused sums)
        M, Cin, Cout, slabs, has_add, fused = args[:6]
        key = ("dgrad+wgrad1x1" + ("+bnsums" if fused else ""), M, 0, Cin, 0, Cout, 1, 1, slabs)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
    elif name == "conv2d_wgrad":
        B, H, W, Cin, Ho, Wo, Cout, K, stride, splits = args[:10]
        if K == 0:                       # the slab reduce of a one-launch 1x1 backward
            key = ("wgrad-reduce", B, H, Cin, Ho, Cout, 1, stride, splits)
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += ms
            continue
        key = ("wgrad", B, H, Cin, Ho, Cout, K, stride, 0)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
print("%-21s %5s %4s %5s %5s %5s %2s %2s %2s %3s %9s %9s %8s %8s" % ("kind", "B", "Hin", "Cin", "Hout", "Cout", "k", "s", "d", "n", "us each", "us total", "TFLOP/s", "TB/s min"))
tot = collections.Counter()
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, B, Hin, Cin, Hout, Nout, K, stride, dil = key
    if kind.startswith("dgrad+wgrad1x1"):      # B column = pixels M; both products; minimum traffic dy + x + dx (+ addsrc, y of the sums)
        fl = 4.0 * B * Cin * Nout
        by = 2.0 * B * (Nout + 2 * Cin + (2 * Cin if kind.endswith("bnsums") else Cin))
    elif kind == "wgrad-reduce":
        fl = 0.0
        by = 4.0 * dil * Cin * Nout
    elif kind == "fwd":
        fl = 2.0 * B * Hout * Hout * Nout * K * K * Cin
        by = 2.0 * B * (Hin * Hin * Cin + Hout * Hout * Nout)
    elif kind == "wgrad":
        fl = 2.0 * B * Hout * Hout * Nout * K * K * Cin
        by = 2.0 * B * (Hin * Hin * Cin + Hout * Hout * Nout)
    else:
        fl = 2.0 * B * Hin * Hin * Cin * K * K * Nout
        by = 2.0 * B * (Hin * Hin * Cin + Hout * Hout * Nout)
    each = ms / n
    tot[kind] += ms
    print("%-21s %5d %4d %5d %5d %5d %2d %2d %2d %3d %9.1f %9.1f %8.0f %8.2f" % (kind, B, Hin, Cin, Hout, Nout, K, stride, dil, n, each * 1e3, ms * 1e3, fl / each / 1e9, by / each / 1e9))
print("totals (ms):", {k: round(v, 3) for k, v in tot.items()})
