"""BASELINE.json configs[2] and configs[3] at their FULL counts on one MI355X (bench.py's encoder figures are a few timed
batches; this runs the whole job, so clock droop over a minute of MFMA work and the allocator's steady state are in it):

  configs[2]  DPR bert-base passage encoder over 1,500,000 synthetic 100-token passages (fresh ids per batch, generated on
              the device before the batch's clock starts; 768-d pooler outputs written into one [1.5M, 768] f32 matrix)
  configs[3]  CLIP ViT-B/32 over 524,288 synthetic 224x224 images (fresh ~N(0,1) pixels per batch), then those 524,288
              512-d vectors as queries, 4096 at a time, top-100 over a 1.5M x 512 "L2norm,Flat" inner-product index

Every encoder output is produced by the product path and kept (nothing cached, nothing skipped); the search leg checks
its first chunk against the exact fp32 scan.  usage: python tools/full_configs.py [passages] [images]  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from bench_encoders import BERT_BASE, CLIP_VITB32, random_bert_state, random_clip_state
from viquae_amd.encoders import CLIPModel, DPRContextEncoder
from viquae_amd.index import MI355XFlatIndex


def dpr_job(n_passages, B=2048, L=100, device="cuda"):
    model = DPRContextEncoder.from_state_dict(dict(BERT_BASE), random_bert_state(BERT_BASE, 0)).to(device).eval()
    out = torch.empty((n_passages, 768), dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    tt = torch.zeros((B, L), dtype=torch.int64, device=device)
    mask = torch.ones((B, L), dtype=torch.int64, device=device)
    model(input_ids=torch.randint(1000, 30000, (B, L), generator=g, device=device), token_type_ids=tt, attention_mask=mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_batch = []
    for i in range(0, n_passages, B):
        b = min(B, n_passages - i)
        ids = torch.randint(1000, 30000, (b, L), generator=g, device=device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out[i:i + b] = model(input_ids=ids, token_type_ids=tt[:b], attention_mask=mask[:b])["pooler_output"]
        torch.cuda.synchronize()
        per_batch.append(time.perf_counter() - t1)
    wall = time.perf_counter() - t0
    enc = sum(per_batch)
    assert bool(torch.isfinite(out).all())
    full = [t for t in per_batch[:-1]] or per_batch
    return {"passages": n_passages, "batch": B, "seq_len": L, "encode_s": enc, "wall_s_with_input_generation": wall,
            "passages_per_s": n_passages / enc, "first_10_batches_ms": 1e3 * sum(full[:10]) / len(full[:10]),
            "last_10_batches_ms": 1e3 * sum(full[-10:]) / len(full[-10:])}, out


def clip_job(n_images, B=3072, device="cuda"):
    model = CLIPModel.from_state_dict(dict(CLIP_VITB32), random_clip_state(CLIP_VITB32, 0)).to(device).eval()
    out = torch.empty((n_images, 512), dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(2)
    model.get_image_features(pixel_values=torch.randn((B, 3, 224, 224), generator=g, device=device))
    torch.cuda.synchronize()
    per_batch = []
    for i in range(0, n_images, B):
        b = min(B, n_images - i)
        px = torch.randn((b, 3, 224, 224), generator=g, device=device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out[i:i + b] = model.get_image_features(pixel_values=px)
        torch.cuda.synchronize()
        per_batch.append(time.perf_counter() - t1)
    enc = sum(per_batch)
    assert bool(torch.isfinite(out).all())
    return {"images": n_images, "batch": B, "encode_s": enc, "images_per_s": n_images / enc}, out


def search_job(queries, n_rows=1_500_000, k=100, chunk=4096, device="cuda"):
    d = queries.shape[1]
    g = torch.Generator(device=device).manual_seed(3)
    X = torch.randn((n_rows, d), generator=g, device=device)
    # SURVEY 8d configs[3]: "L2norm,Flat" + inner product (the reference's CLIP index): rows and queries normalised by the index
    index = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=True)
    index.add(X)
    # CLIP outputs of random pixels through random weights share one large common component: centre them like embeddings of distinct
    # images would be, or every query asks the same question
    q = queries - queries.mean(dim=0, keepdim=True)
    D0, I0 = index.search_device(q[:chunk], k)
    exact = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=False)
    exact.add(X)
    D1, I1 = exact.search_device(q[:chunk], k)
    same = bool(torch.equal(I0, I1) and torch.equal(D0, D1))
    del exact, X
    torch.cuda.synchronize()
    ids = torch.empty((q.shape[0], k), dtype=torch.int64, device=device)
    t0 = time.perf_counter()
    for i in range(0, q.shape[0], chunk):
        ids[i:i + chunk] = index.search_device(q[i:i + chunk], k)[1]
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    st = index.screen_stats(min(chunk, q.shape[0]), k)
    return {"queries": q.shape[0], "rows": n_rows, "d": d, "k": k, "chunk": chunk, "search_s": t, "queries_per_s": q.shape[0] / t,
            "first_chunk_equals_exact_scan": same, "last_chunk_tiles_recomputed": int(st[0])}


def main():
    n_passages = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
    n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 524_288
    rec = {}
    rec["configs2_dpr"], emb = dpr_job(n_passages)
    del emb
    torch.cuda.empty_cache()
    rec["configs3_clip"], feats = clip_job(n_images)
    torch.cuda.empty_cache()
    rec["configs3_search"] = search_job(feats)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
