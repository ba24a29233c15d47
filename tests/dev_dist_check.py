"""Developer script: exercise the nccl (RCCL) code paths with world_size 1 under torchrun on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import openwurli_amd as ow
from openwurli_amd import distributed as owd
rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
jobs = owd.model_notes_job_list()[:40]
t = time.time()
out = owd.batch_render_sharded(jobs, 44100.0, 0.25)
if rank == 0:
    ref = ow.batch_render(jobs, 44100.0, 0.25)
    print("sharded gather ok:", out.shape, float(np.max(np.abs(out - ref.astype(np.float32)))), "t=%.2f" % (time.time() - t))
dist.destroy_process_group()
