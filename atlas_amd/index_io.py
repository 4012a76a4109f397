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
from .passage_store import PassageStore, PassageStoreError

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
        for s in range(opt.save_index_n_shards):       # size AND mtime of every shard pickle: an index re-saved to the same path with equal-sized
            st = os.stat(os.path.join(opt.load_index_path, f"passages.{s}.pt"))        # pickles must not resolve ids to the old text
            parts += [str(st.st_size), str(st.st_mtime_ns)]
    else:
        parts = ["jsonl", str(opt.max_passages)]
        for f in opt.passages:
            st = os.stat(f)
            parts += [os.path.abspath(f), str(st.st_size), str(st.st_mtime_ns)]
    h.update("\n".join(parts).encode())
    return h.hexdigest()


def _passage_store_path(opt, restored: bool):
    """where the node-local passage store of this run lives, or None for the winners-only exchange. Collective when a process group exists
    (one all_gather_object of the host names, at index construction)."""
    import socket
    import tempfile

    explicit = getattr(opt, "passage_store_path", None)       # (an option: the same on every rank)
    env_off = os.environ.get("ATLAS_PASSAGE_STORE", "").lower() in ("off", "0", "none")
    if dist_utils.get_world_size() < 2:
        # one process: an explicit path is honoured (the store then replaces doc_map lookups of nothing -- harmless), nothing automatic
        return None if explicit is None or env_off or str(explicit).lower() in ("off", "none", "") else explicit
    if explicit is not None and str(explicit).lower() in ("off", "none", ""):
        return None
    if explicit is None and not restored and getattr(opt, "use_file_passages", False):
        return None
    # /dev/shm if it has room for the corpus text (containers often mount 64 MiB there), else the temporary directory (a file the page cache
    # serves). Every rank proposes, rank 0's proposal is taken: the ranks must agree on the path. Whatever differs between ranks -- the
    # environment switch, a corpus file one of them cannot see -- travels IN the one collective: the decision is taken from what every rank
    # reported, so no rank can leave before it while the others wait in it.
    import shutil

    base = tempfile.gettempdir()
    try:
        if restored:
            need = sum(os.path.getsize(os.path.join(opt.load_index_path, f"passages.{s}.pt")) for s in range(opt.save_index_n_shards))
        else:
            need = sum(os.path.getsize(f) for f in opt.passages)
        if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 1.5 * need + (64 << 20):
            base = "/dev/shm"
    except OSError:
        pass                                          # (the builder will say what is wrong, on every rank: PassageStoreError)
    reports = dist_utils.all_gather_object((socket.gethostname(), base, env_off))
    if any(off for _, _, off in reports):
        return None
    if explicit is not None:
        return explicit
    if len({h for h, _, _ in reports}) != 1:
        logger.info("ranks on %d hosts: no automatic passage store (set opt.passage_store_path to a node-local path to get one)", len({h for h, _, _ in reports}))
        return None
    # AUTOMATIC stores live in a per-user 0700 directory (PassageStore.private_dir) and are only reused when every file is this user's and
    # nobody else can write to it (open_shared(require_private=True)): every rank unpickles the payload
    return os.path.join(reports[0][1], "atlas_amd_%d" % os.getuid(), "passages_" + _corpus_signature(opt)[:16])


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

    # The node-local passage store: with it search_knn resolves the winners' text locally and is TWO collectives (queries,
    # packed winners) instead of four. Not a reference option, so the default has to be right without one: when a process group of more
    # than one rank runs on ONE host (the target: 8 x MI355X in one node) the store is built -- once, by the node's first rank -- and attached
    # automatically under /dev/shm. `opt.passage_store_path` = an explicit location (multi-node jobs: a node-local path), or "off" to keep
    # the winners-only text exchange (also: ATLAS_PASSAGE_STORE=off).
    store_path = _passage_store_path(opt, restored)
    if store_path and (restored or not opt.use_file_passages):
        if restored:
            def make():
                return PassageStore.iter_saved_index(opt.load_index_path, opt.save_index_n_shards)
        else:
            def make():
                return PassageStore.iter_jsonl(opt.passages, opt.max_passages)
        explicit = getattr(opt, "passage_store_path", None) is not None
        try:
            if not explicit and PassageStore.node_local_rank(getattr(opt, "local_rank", None)) == 0:
                try:
                    PassageStore.private_dir(os.path.dirname(os.path.dirname(store_path)))
                except (OSError, PassageStoreError):
                    pass                                # (open_shared's builder fails on the missing / foreign directory and tells every rank)
            store = PassageStore.open_shared(store_path, make, signature=_corpus_signature(opt), local_rank=getattr(opt, "local_rank", None),
                                             require_private=not explicit)
        except PassageStoreError as e:
            if explicit:
                raise                                   # the caller asked for a store at that path: say so (on every rank alike)
            # the AUTOMATIC store is an optimisation: without it the search keeps the winners-only text exchange (every rank gets here together)
            logger.warning("%s; searching with the winners-only text exchange instead", e)
            return index, passages
        index.attach_passage_store(store)
        if dist_utils.get_rank() == 0:
            # (an automatic store stays under /dev/shm -- RAM -- for the next job on the same corpus, keyed by what it was built from; remove
            #  <path>.bin / .off.npy / .meta.json to free it)
            logger.info("node-local passage store %s: %d passages, %.1f MB", store_path, len(store), os.path.getsize(store_path + ".bin") / 1e6)

    return index, passages
