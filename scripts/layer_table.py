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
    elif name == "conv2d_wgrad":
        B, H, W, Cin, Ho, Wo, Cout, K, stride, splits = args[:10]
        key = ("wgrad", B, H, Cin, Ho, Cout, K, stride, 0)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
print("%-13s %5s %4s %5s %5s %5s %2s %2s %2s %3s %9s %9s %8s %8s" % ("kind", "B", "Hin", "Cin", "Hout", "Cout", "k", "s", "d", "n", "us each", "us total", "TFLOP/s", "TB/s min"))
tot = collections.Counter()
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, B, Hin, Cin, Hout, Nout, K, stride, dil = key
    if kind == "fwd":
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
    print("%-13s %5d %4d %5d %5d %5d %2d %2d %2d %3d %9.1f %9.1f %8.0f %8.2f" % (kind, B, Hin, Cin, Hout, Nout, K, stride, dil, n, each * 1e3, ms * 1e3, fl / each / 1e9, by / each / 1e9))
print("totals (ms):", {k: round(v, 3) for k, v in tot.items()})
