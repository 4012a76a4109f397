"""Index factory + passage loader with the reference's semantics (src/index_io.py:17-93).

`load_or_initialize_index(opt)` is the plug-in point (src/index_io.py:72-93): the reference builds
`DistributedIndex()` for `--index_mode flat`; here the same mode (and the alias "hip") builds
`HipDistributedIndex`. FAISS modes are out of scope and raise the reference's error.
"""
import hashlib
import json
import logging
import os

from . import dist_utils
from .index import HipDistributedIndex
from .passage_store import PassageStore

logger = logging.getLogger(__name__)

_INDEX_CLASSES = {"flat": HipDistributedIndex, "hip": HipDistributedIndex}


def parse_passage_line(line: str):
    """One jsonl line -> passage dict (src/index_io.py:26-34): `id` is mandatory, a non-empty `section` is folded into the
    title; a blank line gives None (the reference keeps the slot: global line numbers must not shift)."""
    if not line.strip():
        return None
    item = json.loads(line)
    assert "id" in item
    if item.get("section") and "title" in item:
        item["title"] = f"{item['title']}: {item['section']}"
    return item


def iter_passage_lines(filenames, maxload=-1):
    """(global line number, raw line) over all files, cut at maxload: the numbering every rank agrees on (index_io.py:36-44)."""
    number = 0
    for filename in filenames:
        with open(filename) as fobj:
            for line in fobj:
                if 0 <= maxload <= number:
                    return
                yield number, line
                number += 1


def load_passages(filenames, maxload=-1):
    """This rank's passages: global line c belongs to rank c % W, in line order (src/index_io.py:17-62)."""
    rank, world = dist_utils.get_rank(), dist_utils.get_world_size()
    mine = []
    for number, line in iter_passage_lines(filenames, maxload):
        if number % world != rank:
            continue
        item = parse_passage_line(line)
        if item is None:
            print("empty line")
        mine.append(item)
    return mine


def save_embeddings_and_index(index, opt) -> None:
    """src/index_io.py:65-69"""
    index.save_index(opt.save_index_path, opt.save_index_n_shards)


def _corpus_signature(opt) -> str:
    """What a node-local passage store must have been built from to be reusable for this run."""
    h = hashlib.sha256()
    if opt.load_index_path is not None:
        parts = ["saved", os.path.abspath(opt.load_index_path), str(opt.save_index_n_shards)]
        parts += [str(os.path.getsize(os.path.join(opt.load_index_path, f"passages.{s}.pt"))) for s in range(opt.save_index_n_shards)]
    else:
        parts = ["jsonl", str(opt.max_passages)]
        for f in opt.passages:
            st = os.stat(f)
            parts += [os.path.abspath(f), str(st.st_size), str(int(st.st_mtime))]
    h.update("\n".join(parts).encode())
    return h.hexdigest()


def load_or_initialize_index(opt):
    """src/index_io.py:72-93 with the flat index replaced by the HIP one: returns (index, this rank's passages)."""
    try:
        index = _INDEX_CLASSES[opt.index_mode]()
    except KeyError:
        raise ValueError(f"unsupported index mode {opt.index_mode}") from None

    restored = opt.load_index_path is not None
    if restored:
        logger.info("restoring the %s index saved under %s (%d shards)", opt.index_mode, opt.load_index_path, opt.save_index_n_shards)
        index.load_index(opt.load_index_path, opt.save_index_n_shards)
        passages = [index.doc_map[row] for row in range(len(index.doc_map))]
    elif opt.use_file_passages:
        passages = []                       # the task files carry their own passages: nothing to index (index_io.py:89)
    else:
        logger.info("reading passages: %s", opt.passages)
        passages = load_passages(opt.passages, opt.max_passages)
        index.init_embeddings(passages)

    # optional, not a reference option: `opt.passage_store_path` = where the node-local passage store lives (e.g. under
    # /dev/shm). With it search_knn resolves the winners' text locally instead of exchanging it.
    store_path = getattr(opt, "passage_store_path", None)
    if store_path and (restored or not opt.use_file_passages):
        if restored:
            def make():
                return PassageStore.iter_saved_index(opt.load_index_path, opt.save_index_n_shards)
        else:
            def make():
                return PassageStore.iter_jsonl(opt.passages, opt.max_passages)
        store = PassageStore.open_shared(store_path, make, signature=_corpus_signature(opt), local_rank=getattr(opt, "local_rank", None))
        index.attach_passage_store(store)

    return index, passages
