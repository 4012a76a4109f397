"""BASELINE configs[4] without the reader: the retrieve-only loop of evaluate.py:40-83 on a saved Atlas index (VERDICT r05 missing #2 / next #7).

    python tools/retrieve_only.py --index DIR --shards 128 --checkpoint $ATLAS_CONTRIEVER_DIR --queries nq-dev.jsonl [--n-context 40] [--batch 64]

The index directory is the reference's on-disk format (embeddings.{i}.pt + passages.{i}.pt, e.g. indices/atlas/wiki/base of the Atlas release, or
one written by tools/refresh_real.py --save-index); queries are jsonl lines with "question" (or "query") and optionally "answers" / "target".
Per batch: `Atlas.retriever_tokenize` -> `Atlas._retrieve` (query embedding in model precision by the HIP encoder + `search_knn` top-k through
the HIP index): end-to-end queries/s, the share spent in search_knn, and -- when answers are given -- answer-string recall@k (lower-cased
containment: a readiness check of the retrieved passages, not the reference's reader EM). One JSON line on stdout.
Multi-GPU: launch with torch.distributed.run; every rank loads its shards (src/index.py:95-99) and takes queries rank::world."""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import real_common  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", required=True)
    ap.add_argument("--shards", type=int, required=True)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--tokenizer", default=None)
    ap.add_argument("--queries", required=True)
    ap.add_argument("--n-context", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--max-queries", type=int, default=-1)
    ap.add_argument("--text-maxlength", type=int, default=200)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    from atlas_amd import index_io

    opt_ix = types.SimpleNamespace(index_mode="flat", load_index_path=args.index, save_index_n_shards=args.shards, passages=[], max_passages=-1,
                                   use_file_passages=False)
    t0 = time.perf_counter()
    index, passages = index_io.load_or_initialize_index(opt_ix)
    t_load = time.perf_counter() - t0
    atlas, opt, which = real_common.make_atlas(args.checkpoint, args.tokenizer, args.text_maxlength)
    rows = [json.loads(ln) for ln in open(args.queries) if ln.strip()]
    if args.max_queries > 0:
        rows = rows[: args.max_queries]
    mine = rows[rank::world]
    n_batches = -(-len(rows[0::world]) // args.batch)                 # search_knn is a collective: every rank runs the same number of batches (evaluate.py:30-35)
    hits = [0, 0]
    t_search = 0.0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for bi in range(n_batches):
        batch = mine[bi * args.batch: (bi + 1) * args.batch]
        query = [r.get("question", r.get("query", "")) for r in batch] or [""]      # a padding batch (evaluate.py:33)
        enc = atlas.retriever_tokenize(query)
        ts = time.perf_counter()
        res = atlas._retrieve(index, args.n_context, query, enc["input_ids"], enc["attention_mask"])
        t_search += time.perf_counter() - ts
        docs = res[0]
        for r, ds in zip(batch, docs):
            answers = r.get("answers") or ([r["target"]] if "target" in r else [])
            if answers:
                text = " ".join((d.get("title", "") + " " + d.get("text", "")).lower() for d in ds)
                hits[0] += int(any(a.lower() in text for a in answers))
                hits[1] += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt, float(hits[0]), float(hits[1])], dtype=torch.float64, device="cuda")
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t)
        dt, hits = float(mx[0]), [int(t[1]), int(t[2])]
    if rank == 0:
        print(json.dumps({"metric": "retrieve-only queries/s (evaluate.py:40-83 without the reader)", "queries": len(rows), "n_gpus": world,
                          "queries_per_s": len(rows) / dt, "seconds": dt, "retrieve_call_share": t_search / dt, "n_context": args.n_context,
                          "batch": args.batch, "index_rows_this_rank": int(index._slab.shape[0]), "index_load_seconds": t_load, "atlas_class": which,
                          "answer_string_recall_at_k": (hits[0] / hits[1]) if hits[1] else None, "answered": hits[1],
                          "search_stats": {k: v for k, v in index.last_search_stats.items() if k in ("path", "fallback_queries", "plan")}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
