import os, sys, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, coracle
from __graft_entry__ import load_package
pkg = load_package()
from test_gpu_sketch import rand_seq
rng = random.Random(77)
n_groups = 23
recs = [rand_seq(rng, rng.choice([0, 5, 20, 21, 22, 100, 151, 2000, 30000]), bad=rng.choice([0, 0, 0.002])) for _ in range(400)]
interleaved = [rng.randrange(n_groups) for _ in recs]
recs, groups = recs[:60], interleaved[:60]
case = (0, 21, True, 42, 1 << 60, True)
gs = [pkg.KmerMinHash(*case) for _ in range(n_groups)]
os_ = [coracle.MinHash(*case) for _ in range(n_groups)]
pkg.KmerMinHash.add_sequences_grouped(gs, recs, groups, True)
for r, g in zip(recs, groups):
    os_[g].add_sequence(r, True)
for gi in range(n_groups):
    if gs[gi].mins != os_[gi].mins or gs[gi].abunds != os_[gi].abunds:
        miss = sorted(set(os_[gi].mins) - set(gs[gi].mins)); extra = sorted(set(gs[gi].mins) - set(os_[gi].mins))
        print("group", gi, "missing", miss[:5], "extra", extra[:5], "records", [(ri, len(recs[ri])) for ri in range(60) if groups[ri] == gi])
        for ri in range(60):
            if groups[ri] != gi: continue
            o = coracle.MinHash(*case); o.add_sequence(recs[ri], True)
            g1 = pkg.KmerMinHash(*case); g1.add_sequence(recs[ri], True)
            if set(o.mins) & set(miss): print("   rec", ri, len(recs[ri]), "holds a missing hash; alone: gpu", g1.mins == o.mins, recs[ri][:40])
# single-sketch over all records
g = pkg.KmerMinHash(*case); o = coracle.MinHash(*case)
g.add_sequences(recs, True)
for r in recs: o.add_sequence(r, True)
print("all records one sketch:", g.mins == o.mins, g.abunds == o.abunds, len(o.mins))
