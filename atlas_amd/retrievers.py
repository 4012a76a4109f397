"""Contriever retriever with the reference's module surface (src/retrievers.py), HIP passage encoder underneath.

What is mirrored (same names, argument meaning, state-dict keys):
    EMBEDDINGS_DIM                       retrievers.py:13
    Contriever(config, pooling="average").forward(input_ids, attention_mask, token_type_ids, ..., normalize=False)
                                         retrievers.py:16-60  (HF BertModel parameter names, so checkpoints
                                         `retriever.contriever.*` load unchanged, model_io.py:62-71)
    BaseRetriever / DualEncoderRetriever / UntiedDualEncoderRetriever   retrievers.py:63-135

What runs on the GPU: every inference forward of the module goes through the C-ABI `atlas_contriever_embed`
(hand-written MFMA GEMMs, fused attention, the reference's non-standard LayerNorm, pooling; token packing so that
padding costs nothing) in the dtype of its parameters:
    fp16   the inference copy `copy.deepcopy(retriever).half().eval()` of `Atlas.build_index` /
           `retrieve_with_rerank` (atlas.py:54-59, 78, 168): every passage embedding of an index refresh
    fp32 / bf16 / fp16   query embedding in model precision (`--precision`, atlas.py:104)
Inference has no eager-PyTorch fallback: without autograd and outside train-mode dropout a forward either runs on the HIP encoder or
raises AtlasHipError (CPU tensors, missing library).

The training step of the reference (`Atlas.forward`, atlas.py:452-465: the retriever in train mode, under autograd, with the
dropout `set_dropout` put on every nn.Dropout) is not part of the accelerated path; it is served by torch operators on the same
parameters (atlas_amd/retriever_train.py) so that the module stays a drop-in under an unchanged training loop. `Contriever.last_path`
("hip" / "autograd") records which way the last forward went.
"""
import collections
import contextlib
import copy
import ctypes
import logging
import os

import torch
import torch.nn as nn

from . import _lib

logger = logging.getLogger(__name__)

EMBEDDINGS_DIM: int = 768
_POOLING = {"average": 0, "sqrt": 1, "cls": 2}        # ATLAS_POOL_* of include/atlas_hip.h


class BertConfigLite:
    """the fields of the HF BertConfig this path reads (facebook/contriever = BERT-base, README.md:267-274)"""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 initializer_range=0.02, pad_token_id=0, pooling=None, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1):
        self.__dict__.update(locals())
        del self.__dict__["self"]
        if pooling is None:              # like an HF BertConfig: no `pooling` attribute unless one was given, so that
            del self.__dict__["pooling"]  # Contriever(config, pooling=...) decides (retrievers.py:19-20)


class _LayerNormParams(nn.Module):        # parameters of modeling_bert.py:94-103 (the arithmetic lives in encoder.hip)
    def __init__(self, hidden_size, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = _LayerNormParams(c.hidden_size, c.layer_norm_eps)
        # persistent buffer of the reference's BertEmbeddings (modeling_bert.py:205): every Atlas checkpoint carries
        # `...embeddings.position_ids`, and src/model_io.py:122 loads with strict=True
        self.register_buffer("position_ids", torch.arange(c.max_position_embeddings).expand((1, -1)))
        self.dropout = nn.Dropout(c.hidden_dropout_prob)       # modeling_bert.py:202 (train mode only: retriever_train.py)


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)
        self.dropout = nn.Dropout(c.attention_probs_dropout_prob)   # modeling_bert.py:267


class _DenseLN(nn.Module):
    def __init__(self, c, in_features):
        super().__init__()
        self.dense = nn.Linear(in_features, c.hidden_size)
        self.LayerNorm = _LayerNormParams(c.hidden_size, c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)       # modeling_bert.py:380, 459


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _DenseLN(c, c.hidden_size)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _DenseLN(c, c.intermediate_size)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])
        self.gradient_checkpointing = False                    # modeling_bert.py:559


_GRAPH_MAX_SLOTS = 16384      # token slots (n x L) up to which embed_into replays a captured hipGraph: the small-batch GEMM configurations
_GRAPH_CACHE = 24             # captured (weights, n, L) shapes kept per module (LRU)


class Contriever(nn.Module):
    def __init__(self, config=None, pooling="average", **kwargs):
        super().__init__()
        self.config = config or BertConfigLite()
        if not hasattr(self.config, "pooling"):
            self.config.pooling = pooling
        self.embeddings = _Embeddings(self.config)
        self.encoder = _Encoder(self.config)
        self._packed = None          # (key, BertWeights struct, tensors kept alive)
        self._ws = None
        # hipGraph replay of small (query-like) batches, round 6: (weights, n, L) -> captured launch sequence + its static buffers (embed_into)
        self._graphs = collections.OrderedDict()
        self._graph_last_key = None
        self.query_graphs = os.environ.get("ATLAS_QUERY_GRAPHS", "1") != "0"
        self._library = None         # tests / tools only: a handle of the tuning build (_lib.lib(tuning=True)) instead of the product library
        self.last_path = None        # "hip" | "autograd": which implementation served the last forward

    @classmethod
    def from_pretrained(cls, path, pooling="average", **kwargs):
        """`Contriever.from_pretrained(opt.retriever_model_path)` (src/model_io.py:45) for a LOCAL directory in the HF layout:
        config.json + model.safetensors or pytorch_model.bin (facebook/contriever ships both). Keys may carry the `bert.`
        prefix of task heads; `pooler.*` (unused: add_pooling_layer=False, retrievers.py:17) is dropped. No hub download (there is no network on the target boxes)."""
        import json
        import os

        with open(os.path.join(path, "config.json")) as f:
            cj = json.load(f)
        known = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "max_position_embeddings",
                 "type_vocab_size", "layer_norm_eps", "initializer_range", "pad_token_id", "hidden_dropout_prob",
                 "attention_probs_dropout_prob")
        config = BertConfigLite(**{k: cj[k] for k in known if k in cj}, pooling=cj.get("pooling", pooling))
        if cj.get("hidden_act", "gelu") != "gelu":
            raise _lib.AtlasHipError(f"hidden_act={cj['hidden_act']!r}: only the exact-erf 'gelu' of BERT / Contriever is implemented")
        st_path, bin_path = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st_path):
            from safetensors.torch import load_file

            sd = load_file(st_path)
        elif os.path.exists(bin_path):
            sd = torch.load(bin_path, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        clean = {}
        for k, v in sd.items():
            k = k[5:] if k.startswith("bert.") else k
            if k.startswith("pooler."):
                continue
            clean[k] = v
        model = cls(config)
        # checkpoints written by transformers >= 4.31 no longer carry the position_ids buffer
        clean.setdefault("embeddings.position_ids", model.embeddings.position_ids)
        model.load_state_dict(clean, strict=True)
        return model

    # ---- weights -> C-ABI struct (fused QKV), cached until a parameter changes ----
    def _params(self):
        """The module's Parameter objects, enumerated ONCE: `list(self.parameters())` walks ~100 sub-modules through python generators -- 260 us per
        call, three calls per embed, against ~0.9 ms of GPU time for a 64-query batch (round 6: the host-side share of a query embedding was
        0.5 ms). Parameter OBJECTS are stable under everything atlas does to a retriever (`.half()` / `.to()` / `.cuda()` swap `.data` in place,
        `load_state_dict` copies in place, optimisers update in place, `copy.deepcopy` builds a new module: `__deepcopy__` drops the cache);
        `_apply` and `load_state_dict` drop it anyway (the overwrite-on-conversion future flag, `assign=True`), and a caller that REPLACES a
        Parameter object by assignment calls `invalidate_parameter_cache()`."""
        ps = self.__dict__.get("_param_cache")
        if ps is None:
            ps = self.__dict__["_param_cache"] = list(self.parameters())
        return ps

    def invalidate_parameter_cache(self):
        self.__dict__["_param_cache"] = None
        self._packed = None

    def _apply(self, fn, *a, **kw):
        self.__dict__["_param_cache"] = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self.__dict__["_param_cache"] = None
        return out

    def _pack(self):
        params = self._params()
        key = tuple((p.data_ptr(), p._version) for p in params) + (self.config.pooling,)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        c = self.config
        keep = []

        def dev(t):
            t = t.detach().contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.BertWeights()
        w.n_layers, w.n_heads, w.hidden, w.intermediate = c.num_hidden_layers, c.num_attention_heads, c.hidden_size, c.intermediate_size
        w.eps = float(c.layer_norm_eps)
        w.dtype = _lib.torch_dtype_code(self.embeddings.word_embeddings.weight.dtype)
        w.pooling = _POOLING[c.pooling]
        e0 = self.embeddings
        w.vocab_size, w.max_positions, w.type_vocab = (e0.word_embeddings.weight.shape[0], e0.position_embeddings.weight.shape[0],
                                                       e0.token_type_embeddings.weight.shape[0])
        e = self.embeddings
        w.word_emb, w.pos_emb, w.type_emb = dev(e.word_embeddings.weight), dev(e.position_embeddings.weight), dev(e.token_type_embeddings.weight)
        w.emb_ln_w, w.emb_ln_b = dev(e.LayerNorm.weight), dev(e.LayerNorm.bias)
        for i, ly in enumerate(self.encoder.layer):
            s = ly.attention.self
            lw = w.layers[i]
            lw.qkv_w = dev(torch.cat([s.query.weight, s.key.weight, s.value.weight], dim=0))
            lw.qkv_b = dev(torch.cat([s.query.bias, s.key.bias, s.value.bias], dim=0))
            lw.o_w, lw.o_b = dev(ly.attention.output.dense.weight), dev(ly.attention.output.dense.bias)
            lw.ln1_w, lw.ln1_b = dev(ly.attention.output.LayerNorm.weight), dev(ly.attention.output.LayerNorm.bias)
            lw.ff1_w, lw.ff1_b = dev(ly.intermediate.dense.weight), dev(ly.intermediate.dense.bias)
            lw.ff2_w, lw.ff2_b = dev(ly.output.dense.weight), dev(ly.output.dense.bias)
            lw.ln2_w, lw.ln2_b = dev(ly.output.LayerNorm.weight), dev(ly.output.LayerNorm.bias)
        self._packed = (key, w, keep)
        return w

    def _check_accelerated(self):
        params = self._params()
        p = params[0]
        if not p.is_cuda:
            raise _lib.AtlasHipError("atlas_amd.Contriever runs on an MI355X only; there is no CPU / eager fallback")
        if p.dtype not in (torch.float16, torch.bfloat16, torch.float32) or any(q.dtype != p.dtype for q in params):
            raise _lib.AtlasHipError(f"unsupported / mixed parameter dtype {p.dtype}")
        if self._needs_training_forward():
            raise _lib.AtlasHipError("embed_into is the inference encoder: call it under torch.no_grad() on a module in eval mode "
                                     "(a forward that needs autograd or train-mode dropout goes through Contriever.forward)")
        if self.config.pooling not in _POOLING:
            raise _lib.AtlasHipError(f"pooling={self.config.pooling!r}: the reference knows 'average', 'sqrt', 'cls' (retrievers.py:51-56)")
        return p.dtype

    def _needs_training_forward(self) -> bool:
        """autograd through the parameters, or train-mode dropout: the two things the eval-mode HIP encoder does not do"""
        if torch.is_grad_enabled() and any(q.requires_grad for q in self._params()):
            return True
        return self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules())

    def _out_dtype(self):
        # 'sqrt' divides a model-dtype tensor by an fp32 tensor: torch promotes, the reference returns fp32 (retrievers.py:53-54)
        return torch.float32 if self.config.pooling == "sqrt" else self.embeddings.word_embeddings.weight.dtype

    def embed_into(self, out: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids=None,
                   trim_padding=False, out_rows: torch.Tensor = None):
        """Encode a batch and write the (n, 768) embeddings (model dtype) into `out` (contiguous rows; for the
        fp16 copy it may be a slice of the index slab, which fuses atlas.py:79 into the pooling epilogue).
        With `out_rows` (int64 [n] on the device) `out` is the whole (N, 768) destination and passage b lands in row out_rows[b].

        Only unmasked tokens are computed (packed on the device, no host sync). trim_padding=True additionally cuts
        the all-padding tail columns first (one host sync): the launch grids and the GEMM tile shape are sized by
        n*L, which for queries tokenised with padding='max_length' is ~20x the real token count."""
        dtype = self._check_accelerated()
        L = self._library or _lib.lib()
        n, seq = input_ids.shape
        assert out.dtype == self._out_dtype() and out.is_contiguous() and out.shape[1] == EMBEDDINGS_DIM
        if out_rows is None:
            assert out.shape[0] == n
        else:
            assert out_rows.dtype == torch.int64 and out_rows.is_cuda and out_rows.numel() == n and out_rows.is_contiguous()
        if n == 0:
            return out
        if trim_padding:
            used = (attention_mask != 0).any(dim=0).nonzero()
            seq = int(used.max()) + 1 if used.numel() else 1
            seq = min(int(input_ids.shape[1]), (seq + 7) // 8 * 8)      # (a multiple of 8: fewer distinct launch shapes; masked columns change no bit)
            input_ids, attention_mask = input_ids[:, :seq], attention_mask[:, :seq]
            token_type_ids = token_type_ids[:, :seq] if token_type_ids is not None else None
        ids = input_ids.to(torch.int64).contiguous()
        mask = attention_mask.to(torch.int64).contiguous()
        tt = token_type_ids.to(torch.int64).contiguous() if token_type_ids is not None else None
        w = self._pack()
        if (self.query_graphs and out_rows is None and n * seq <= _GRAPH_MAX_SLOTS and not torch.cuda.is_current_stream_capturing()
                and self._embed_graphed(L, w, out, ids, mask, tt, n, seq)):
            return out
        need = L.atlas_contriever_workspace_bytes(n, seq, w.dtype)
        if self._ws is None or self._ws.numel() < need or self._ws.device != ids.device:
            self._ws = None
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=ids.device)
        self._launch(L, w, out, ids, mask, tt, n, seq, out_rows, self._ws)
        # the kernel wrote through a raw pointer: tell torch (an empty in-place op bumps the version counter that `out` shares with the
        # tensor it is a view of -- HipDistributedIndex trusts its measured row-norm bound only while that counter stands still)
        out[:0].zero_()
        return out

    def _launch(self, L, w, out, ids, mask, tt, n, seq, out_rows, ws):
        stream = torch.cuda.current_stream(ids.device).cuda_stream
        _lib.check(L.atlas_contriever_embed_rows(ctypes.byref(w), ids.data_ptr(), mask.data_ptr(), tt.data_ptr() if tt is not None else None,
                                                 n, seq, out.data_ptr(), out_rows.data_ptr() if out_rows is not None else None,
                                                 ws.data_ptr(), ws.numel(), stream),
                   "atlas_contriever_embed")

    def _embed_graphed(self, L, w, out, ids, mask, tt, n, seq) -> bool:
        """Small batches -- the query embedding of src/atlas.py:104: 64 queries x ~20 tokens -- are 87 launches of 3-19 us each: launch-bound.
        The second consecutive call with the same (weights, n, L) captures the launch sequence of the C-ABI call into a hipGraph over static
        input / output / workspace buffers; from then on a call is: two small copies in, one graph launch, one copy out. What it buys, measured
        (profiles/r06/enc_query_time.txt, 64 queries x 23 tokens): against the plain launches of the SAME build 0-5 % (fp16 1.16 -> 1.11 ms with
        trim_padding, 1.01 -> 1.00 without; fp32 none) -- the 0.3-0.5 ms the first probe credited to the graph (enc_query_graph_probe.txt) was the
        python side of the call, which `_params()` removed for both paths. Outputs identical. The key holds the packed weights'
        identity (pointers + torch version counters), so a graph is never replayed over weights that have changed (a training loop, whose
        weights change every step, never sees the same key twice and stays eager). Single-threaded like everything behind this boundary
        (SURVEY §8b): the static buffers belong to the module. Returns False when the call should take the eager path (first sight of a
        key, capture not possible); ATLAS_QUERY_GRAPHS=0 / `module.query_graphs = False` switches it off."""
        key = (self._packed[0], id(L), n, seq, tt is not None, ids.device.index)
        ent = self._graphs.get(key)
        if ent is None:
            seen_before, self._graph_last_key = (self._graph_last_key == key), key
            if not seen_before:
                return False
            try:
                dev = ids.device
                st_ids, st_mask = torch.empty_like(ids), torch.empty_like(mask)
                st_tt = torch.empty_like(tt) if tt is not None else None
                st_out = torch.empty((n, EMBEDDINGS_DIM), dtype=out.dtype, device=dev)
                ws = torch.empty(int(L.atlas_contriever_workspace_bytes(n, seq, w.dtype)), dtype=torch.uint8, device=dev)
                st_ids.copy_(ids); st_mask.copy_(mask)
                if tt is not None:
                    st_tt.copy_(tt)
                graph = torch.cuda.CUDAGraph()
                # (thread-local capture mode: a process-group watchdog thread polling its events -- a multi-rank job -- must neither be
                #  failed by this capture nor invalidate it; unsafe calls from THIS thread still abort the capture and land in `except`)
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._launch(L, w, st_out, st_ids, st_mask, st_tt, n, seq, None, ws)
                ent = (graph, st_ids, st_mask, st_tt, st_out, ws, self._packed[2])       # (the packed weight tensors stay alive with the graph)
            except Exception as e:                                   # noqa: BLE001  (capture is an optimisation: the eager HIP path serves the call)
                logger.warning("hipGraph capture of the query embedding failed (%s: %s); staying on plain launches", type(e).__name__, e)
                self.query_graphs = False
                return False
            self._graphs[key] = ent
            while len(self._graphs) > _GRAPH_CACHE:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        graph, st_ids, st_mask, st_tt, st_out, _, _ = ent
        st_ids.copy_(ids); st_mask.copy_(mask)
        if tt is not None:
            st_tt.copy_(tt)
        graph.replay()
        out.copy_(st_out)                                            # (an in-place torch write: bumps the version counter by itself)
        return True

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, output_attentions=None,
                output_hidden_states=None, normalize=False, trim_padding=False):
        """retrievers.py:22-60. Only the arguments atlas.py passes are supported (ids, mask, token_type_ids)."""
        assert position_ids is None and head_mask is None and inputs_embeds is None and encoder_hidden_states is None
        if self._needs_training_forward():
            from .retriever_train import training_forward

            self.last_path = "autograd"
            return training_forward(self, input_ids, attention_mask, token_type_ids, normalize=normalize)
        self.last_path = "hip"
        out = torch.empty((input_ids.shape[0], EMBEDDINGS_DIM), dtype=self._out_dtype(), device=input_ids.device)
        self.embed_into(out, input_ids, attention_mask, token_type_ids, trim_padding=trim_padding)
        if normalize:
            out = torch.nn.functional.normalize(out, dim=-1).clone()
        return out

    def gradient_checkpointing_enable(self):     # retrievers.py:81-87 calls these on children; the flag of modeling_bert.py:559, 586
        self.encoder.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.encoder.gradient_checkpointing = False

    def __deepcopy__(self, memo):                # atlas.py:59 deep-copies the retriever: drop the packed cache
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        nn.Module.__init__(new)
        for k, v in self.__dict__.items():
            if k in ("_packed", "_ws", "_library", "last_path", "_graph_last_key", "_param_cache"):
                new.__dict__[k] = None
            elif k == "_graphs":
                new.__dict__[k] = collections.OrderedDict()
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


@contextlib.contextmanager
def _frozen_eval(module):
    """run `module` in eval mode without autograd and put its mode back afterwards"""
    was_training = module.training
    module.eval()
    try:
        with torch.no_grad():
            yield module
    finally:
        module.train(was_training)


class BaseRetriever(nn.Module):
    """The retriever surface atlas.py talks to (src/retrievers.py:63-87): `forward(*args, is_passages=False, **kw)` dispatches to
    `embed_passages` / `embed_queries`, and gradient checkpointing is switched on every child encoder.

    Subclasses name the child modules that play the two roles (the attribute names are the checkpoint prefixes
    `retriever.contriever.*` / `retriever.{query,passage}_contriever.*`, src/model_io.py:62-71); a subclass without roles has to
    override the two `embed_*` methods, as in the reference."""

    query_role = None        # attribute holding the query encoder
    passage_role = None      # attribute holding the passage encoder

    def __init__(self, *args, **kwargs):
        super().__init__()

    def _encode(self, role, args, kwargs, queries=False):
        if role is None:
            raise NotImplementedError()
        encoder = getattr(self, role)
        if queries and isinstance(encoder, Contriever):
            # queries arrive padded to max_length (atlas.py retriever_tokenize): size the launch by the real tokens
            kwargs.setdefault("trim_padding", True)
        return encoder(*args, **kwargs)

    def embed_queries(self, *args, **kwargs):
        return self._encode(self.query_role, args, kwargs, queries=True)

    def embed_passages(self, *args, **kwargs):
        return self._encode(self.passage_role, args, kwargs)

    def forward(self, *args, is_passages=False, **kwargs):
        embed = self.embed_passages if is_passages else self.embed_queries
        return embed(*args, **kwargs)

    def _each_child(self, method):
        for child in self.children():
            getattr(child, method)()

    def gradient_checkpointing_enable(self):
        self._each_child("gradient_checkpointing_enable")

    def gradient_checkpointing_disable(self):
        self._each_child("gradient_checkpointing_disable")


class DualEncoderRetriever(BaseRetriever):
    """One shared encoder for queries and passages (src/retrievers.py:90-105)."""

    query_role = passage_role = "contriever"

    def __init__(self, opt, contriever):
        super().__init__()
        self.opt = opt
        self.contriever = contriever

    def _embed(self, *args, **kwargs):       # kept: the reference exposes it
        return self.contriever(*args, **kwargs)


class UntiedDualEncoderRetriever(BaseRetriever):
    """Separate query and passage encoders (src/retrievers.py:108-135). With `opt.query_side_retriever_training` the passage
    encoder is frozen: passages are embedded in eval mode without autograd, whatever mode the module is in."""

    query_role, passage_role = "query_contriever", "passage_contriever"

    def __init__(self, opt, query_encoder, passage_encoder=None):
        super().__init__()
        self.opt = opt
        self.query_contriever = query_encoder
        if passage_encoder is None:
            # as the reference: a wrapped (`.module`) query encoder is copied, a bare one is shared
            passage_encoder = copy.deepcopy(query_encoder) if hasattr(query_encoder, "module") else query_encoder
        self.passage_contriever = passage_encoder

    def embed_passages(self, *args, **kwargs):
        if not self.opt.query_side_retriever_training:
            return super().embed_passages(*args, **kwargs)
        with _frozen_eval(self.passage_contriever):
            return super().embed_passages(*args, **kwargs)
