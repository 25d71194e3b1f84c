"""How many host threads should the CPU-oracle baseline use on this box?  (torch's default = all cores was 40x slower
than 8 threads on a 256-core host.)  Times oracle.generator_forward on a 256x256 generator for several thread counts."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import seeded
from oracle import e4s_oracle as O

man = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "manifest.json")))["generator_256_rli13"]
tmpl = {k: torch.empty(tuple(s), dtype=getattr(torch, d), device="meta") for k, (s, d) in man.items()}
sd = seeded.seeded_state_dict(tmpl, 21, "net3")
codes = seeded.seeded_codes(23, 1, 12, 14, seeded.seeded_latent_avg(2, 14))
mask = seeded.labels_to_onehot(seeded.blocky_labels(22, 1, 12, 64, 8), 12)
print("cpu_count", os.cpu_count(), flush=True)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    with torch.no_grad():
        t = time.perf_counter(); O.generator_forward(sd, codes, mask, None, size=256); dt = time.perf_counter() - t
    print(f"threads={nt:4d}  gen256: {dt:.2f}s", flush=True)
