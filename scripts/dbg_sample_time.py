import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tortoise_tts_amd import engine as E, stages, weights as W
from tortoise_tts_amd.config import ARConfig
from bench import bench_prompt
import torch.nn.functional as F
lib = E.init()
cfg = ARConfig()
sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
ar = stages.ArStage(sd, cfg, max_batch=256, max_new_tokens=64, max_latent_candidates=1)
text, (auto, _) = bench_prompt()
tt = F.pad(text.int()[None], (0, 1)).cuda()
lib.tt_graph_replay(0); lib.tt_prof_enable(1)
ar.prefill(auto.cuda(), tt)
ar.generate(256, 30, seed=1)
torch.cuda.synchronize(); lib.tt_prof_enable(0)
buf = (C.c_double * 4)()
for i in range(lib.tt_prof_classes()):
    lib.tt_prof_read(i, buf)
    if buf[0] > 0 and b"sample" in lib.tt_prof_class_name(i):
        print("stop=%s sample_kernel %.2f us" % (os.environ.get("AB_TAG"), 1e3 * buf[1] / buf[0]))
