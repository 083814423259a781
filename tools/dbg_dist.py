import os, sys, random, threading
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
from flow2gan_amd import dist as fdist
import test_zz_hip_dist as Tz
dist.init_process_group("gloo", rank=0, world_size=1)
gan, logmel = Tz._build()
red = fdist.GradReducer(bucket_mb=8.0, force=True)
names = {id(p): n for n, p in gan.named_parameters()}
orig = fdist.GradReducer._on_grad
IN_DELIVER = [False]
od = fdist._Sink.deliver
def deliver(self, key, params, grads):
    IN_DELIVER[0] = True
    print("DELIVER", names[id(params[0])], "uses", self.uses[key], [g is not None for g in grads], flush=True)
    try:
        return od(self, key, params, grads)
    finally:
        IN_DELIVER[0] = False
fdist._Sink.deliver = deliver
def spy(self, p):
    plan = self._active
    b = plan.bucket_of.get(id(p)) if plan else None
    if b is not None and "discriminators.0.convs" in names[id(p)] and names[id(p)].startswith("discriminator.0."):
        print("ONGRAD", names[id(p)], "from_deliver", IN_DELIVER[0], "sent", b.sent, "bucket", b.index,
              "grad is view", p.grad.data_ptr() == plan.view_of[id(p)].data_ptr(), threading.current_thread().name, flush=True)
    return orig(self, p)
fdist.GradReducer._on_grad = spy
try:
    Tz._steps(gan, logmel, 0, red)
    print("OK")
except Exception as e:
    print("ERR", e)
