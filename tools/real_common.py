"""shared by tools/refresh_real.py and tools/retrieve_only.py: an Atlas-shaped holder of (retriever, tokenizer, opt) for the refresh / retrieval
calls of src/atlas.py, with the reference's own `Atlas` class when a checkout is reachable ($ATLAS_REFERENCE_DIR, /root/reference, .refstage)
and a line-by-line restatement of the three methods used here otherwise (the reference's sources do not travel to the GPU box)."""
import importlib
import math
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BERT_MAX_SEQ_LENGTH = 512                                     # src/atlas.py:25


def reference_dir():
    for cand in (os.environ.get("ATLAS_REFERENCE_DIR"), "/root/reference", os.path.join(ROOT, ".refstage")):
        if cand and os.path.exists(os.path.join(cand, "src", "atlas.py")):
            return cand
    return None


class AtlasRestated(torch.nn.Module):
    """src/atlas.py:54-118, 184-198 restated (build_index, _retrieve, retriever_tokenize): used only when no reference checkout is reachable"""

    def __init__(self, opt, retriever, retriever_tokenizer):
        super().__init__()
        self.opt, self.retriever, self.retriever_tokenizer = opt, retriever, retriever_tokenizer

    def _get_fp16_retriever_copy(self):                      # atlas.py:54-59
        import copy

        return copy.deepcopy(self.retriever.module if hasattr(self.retriever, "module") else self.retriever).half().eval()

    @torch.no_grad()
    def build_index(self, index, passages, gpu_embedder_batch_size, logger=None):      # atlas.py:61-88
        n_batch = math.ceil(len(passages) / gpu_embedder_batch_size)
        retrieverfp16 = self._get_fp16_retriever_copy()
        total = 0
        for i in range(n_batch):
            batch = passages[i * gpu_embedder_batch_size: (i + 1) * gpu_embedder_batch_size]
            batch = [self.opt.retriever_format.format(**example) for example in batch]
            batch_enc = self.retriever_tokenizer(batch, padding="longest", return_tensors="pt",
                                                 max_length=min(self.opt.text_maxlength, gpu_embedder_batch_size), truncation=True)
            embeddings = retrieverfp16(**{k: v.cuda() for k, v in batch_enc.items()}, is_passages=True)
            index.embeddings[:, total: total + len(embeddings)] = embeddings.T
            total += len(embeddings)

    def retriever_tokenize(self, query):                     # atlas.py:184-198
        enc = self.retriever_tokenizer(query, max_length=min(self.opt.text_maxlength, BERT_MAX_SEQ_LENGTH), padding="max_length", truncation=True,
                                       return_tensors="pt")
        return {k: v.cuda() for k, v in enc.items()}

    @torch.no_grad()
    def _retrieve(self, index, topk, query, query_ids_retriever, query_mask_retriever):    # atlas.py:90-118 without a task filter
        self.retriever.eval()
        query_emb = self.retriever(query_ids_retriever, query_mask_retriever, is_passages=False)
        t = time.time()
        passages, scores = index.search_knn(query_emb, topk)
        return passages, scores, query_emb, time.time() - t


def make_atlas(checkpoint, tokenizer_dir=None, text_maxlength=200, retriever_format="{title} {text}", precision=torch.float32):
    """(atlas-like object, 'reference' | 'restated'): the HIP Contriever from `checkpoint` inside DualEncoderRetriever (src/model_io.py:41-59),
    the HF tokenizer of the checkpoint directory"""
    import transformers

    from atlas_amd import retrievers as R

    opt = types.SimpleNamespace(retriever_format=retriever_format, text_maxlength=text_maxlength, filtering_overretrieve_ratio=2,
                                n_to_rerank_with_retrieve_with_rerank=128, retrieve_with_rerank=False, query_side_retriever_training=False)
    encoder = R.Contriever.from_pretrained(checkpoint)
    retriever = R.DualEncoderRetriever(opt, encoder).to(precision).cuda()
    tokenizer = transformers.AutoTokenizer.from_pretrained(tokenizer_dir or checkpoint)
    ref = reference_dir()
    if ref is not None:
        for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
            del sys.modules[name]
        sys.path.insert(0, ref)
        sys.modules["src.retrievers"] = R
        try:
            mod = importlib.import_module("src.atlas")
            atlas = mod.Atlas(opt, torch.nn.Linear(1, 1), retriever, types.SimpleNamespace(vocab={"a": 0}), tokenizer)
            return atlas, opt, "reference src/atlas.py (%s)" % ref
        except Exception as e:                               # noqa: BLE001  (e.g. a dependency of atlas.py that is not installed)
            print(f"reference Atlas class not usable ({type(e).__name__}: {e}); using the restated methods", file=sys.stderr)
        finally:
            sys.path.remove(ref)
    return AtlasRestated(opt, retriever, tokenizer), opt, "restated (tools/real_common.py)"
