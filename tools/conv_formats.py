"""Dev experiment: every distinct 3x3 convolution of the U-net at 1280x720, forward + backward,
NCHW vs channels_last activations under MIOPEN_FIND_MODE=FAST (what bench.py uses)."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import modules

dev = th.device("cuda")
net = modules.Autoencoder(128, 128, num_levels=3, increase_factor=2.0, num_convs=3, width=128, ksize=3,
                          output_type="leaky_relu", pooling="max").to(dev)
shapes = []
def hook(m, inp, out):
    shapes.append((m.in_channels, m.out_channels, inp[0].shape[-2], inp[0].shape[-1]))
hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, th.nn.Conv2d)]
with th.no_grad():
    net(th.randn(1, 128, 720, 1280, device=dev))
for h in hs:
    h.remove()
del net
th.cuda.empty_cache()
uniq = sorted(set(shapes), key=shapes.index)
print("convs per U-net:", len(shapes), "distinct:", len(uniq), flush=True)

def bench(cin, cout, h, w, cl):
    conv = th.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = th.randn(1, cin, h, w, device=dev)
    if cl:
        conv = conv.to(memory_format=th.channels_last)
        x = x.contiguous(memory_format=th.channels_last)
    x.requires_grad_()
    def step():
        y = conv(x)
        y.backward(y.detach())
    t0 = time.time()
    step(); th.cuda.synchronize()
    first = time.time() - t0          # includes MIOpen's solver search for this configuration
    step(); th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3, first

tot = [0.0, 0.0]
only_cl = "--only-channels-last" in sys.argv
for (cin, cout, h, w) in uniq:
    n = shapes.count((cin, cout, h, w))
    a, fa = (0.0, 0.0) if only_cl else bench(cin, cout, h, w, False)
    b, fb = bench(cin, cout, h, w, True)
    tot[0] += a * n; tot[1] += b * n
    print("%4d->%4d %4dx%4d x%d: NCHW %.2f ms (first %.1fs) | channels_last %.2f ms (first %.1fs)" % (
        cin, cout, h, w, n, a, fa, b, fb), flush=True)
print("per U-net: NCHW %.1f ms, channels_last %.1f ms" % tuple(tot))
