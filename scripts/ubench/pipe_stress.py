#!/usr/bin/env python3
"""stress of the in-flight paths: many rounds of K forwards through PipelinedVAD / ShardedPipeline (without a process group, then through
RCCL with one rank, both gather modes), every output compared with the module's own result for the same input.  pipe_stress.py [rounds]"""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice_activity_detection_amd import PipelinedVAD, SelfAttentiveVAD, seeded_features, seeded_state_dict
from voice_activity_detection_amd.distributed import ShardedPipeline

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()

def run(tag, shape, make, rounds):
    xs = [torch.from_numpy(seeded_features(10 * i + shape[1], shape)).cuda() for i in range(4)]
    with torch.no_grad():
        want = [m(features=x).clone() for x in xs]
    torch.cuda.synchronize()
    sp = make()
    bad, worst = 0, 0.0
    for r in range(rounds):
        if isinstance(sp, PipelinedVAD):
            outs = [o.clone() for o in sp.forward_many([x * 1.0 for x in xs])]
        else:
            for x in xs:
                sp.submit(x * 1.0)
            outs = [o[0].clone() for o in sp.join()]
        for o, w in zip(outs, want):
            if not torch.equal(o, w):
                bad += 1
                worst = max(worst, float((o - w).abs().max()))
    torch.cuda.synchronize()
    print(f"{tag:34s} {shape}: {bad} of {rounds * 4} outputs differ from the module's (worst {worst:.2e})", flush=True)

for shape in ((8, 200, 80), (5, 96, 80), (16, 300, 80)):
    run("PipelinedVAD depth 3", shape, lambda: PipelinedVAD(m, 3), rounds)
    run("ShardedPipeline, no process group", shape, lambda: ShardedPipeline(m, slots=4, depth=3, gather="step"), rounds)
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for shape in ((8, 200, 80), (5, 96, 80)):
    for g in ("step", "final"):
        run(f"ShardedPipeline, RCCL world 1, {g}", shape, lambda: ShardedPipeline(m, slots=4, depth=3, gather=g), rounds)
torch.cuda.synchronize(); dist.destroy_process_group()
