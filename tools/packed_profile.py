"""Only the packed DPR forward on the reference's padded batches (2048 x pad-to-256, lengths ~N(130,30)): for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench_encoders import BERT_BASE, random_bert_state, time_it
from viquae_amd.encoders import DPRContextEncoder
B, L = 2048, 256
model = DPRContextEncoder.from_state_dict(dict(BERT_BASE), random_bert_state(BERT_BASE, 0)).to("cuda").eval()
rng = np.random.default_rng(4)
lens = np.clip(rng.normal(130, 30, B).astype(int), 8, L)
mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64)).cuda()
ids = torch.randint(1000, 30000, (B, L), device="cuda") * mask
run = lambda: model(input_ids=ids, attention_mask=mask)["pooler_output"]
run()
t = time_it(run, 3)
print(f"packed: {t * 1e3:.1f} ms per {B} passages ({B / t:.0f} passages/s, {int(lens.sum())} real tokens, {t / lens.sum() * 1e6:.3f} us/token)")
