"""What a plain streaming kernel reaches on 74 MB from cold caches (the state the STFT of a train step finds): a float4 elementwise
pass reading 37 MB and writing 37 MB after a 2 GiB fill, vs back to back."""
import torch
dev = "cuda"
n = 37 * 1000 * 1000 // 4
a = torch.randn(n, device=dev); b = torch.empty_like(a)
junk = torch.empty(1 << 29, device=dev)
def run(): torch.mul(a, 2.0, out=b)
run(); torch.cuda.synchronize()
cold = []
for it in range(12):
    junk.fill_(float(it))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    cold.append(e0.elapsed_time(e1) * 1e3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
cold.sort()
nb = 2 * n * 4
print("elementwise 37 MB -> 37 MB: cold %.1f us (%.2f TB/s), back to back %.1f us (%.2f TB/s)" % (cold[6], nb / cold[6] / 1e6, e0.elapsed_time(e1) / 50 * 1e3, nb / (e0.elapsed_time(e1) / 50 * 1e3) / 1e6))
# an empty launch, for the fixed cost inside an event pair
z = torch.empty(1, device=dev)
ts = []
for it in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); z.fill_(1.0); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("one-element fill between two events: %.1f us" % ts[10])
