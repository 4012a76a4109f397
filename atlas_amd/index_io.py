"""Index factory + passage loader with the reference's semantics (src/index_io.py:17-93).

`load_or_initialize_index(opt)` is the plug-in point (src/index_io.py:72-93): the reference builds
`DistributedIndex()` for `--index_mode flat`; here the same mode (and the alias "hip") builds
`HipDistributedIndex`. FAISS modes are out of scope and raise the reference's error.
"""
import json
import logging

from . import dist_utils
from .index import HipDistributedIndex
from .passage_store import PassageStore

logger = logging.getLogger(__name__)


def load_passages(filenames, maxload=-1):
    """jsonl passages, round-robin by global line number over ranks (src/index_io.py:17-62)."""
    counter = 0
    passages = []
    global_rank = dist_utils.get_rank()
    world_size = dist_utils.get_world_size()
    for filename in filenames:
        with open(filename) as fobj:
            for line in fobj:
                if maxload > -1 and counter >= maxload:
                    break
                if (counter % world_size) == global_rank:
                    if line.strip() != "":
                        item = json.loads(line)
                        assert "id" in item
                        if "title" in item and "section" in item and len(item["section"]) > 0:
                            item["title"] = f"{item['title']}: {item['section']}"
                        passages.append(item)
                    else:
                        print("empty line")
                        passages.append(None)   # the reference appends None for blank lines too
                counter += 1
    return passages


def save_embeddings_and_index(index, opt) -> None:
    """src/index_io.py:65-69"""
    index.save_index(opt.save_index_path, opt.save_index_n_shards)


def load_or_initialize_index(opt):
    """src/index_io.py:72-93 with the flat index replaced by the HIP one."""
    if opt.index_mode in ("flat", "hip"):
        index = HipDistributedIndex()
    else:
        raise ValueError(f"unsupported index mode {opt.index_mode}")

    if opt.load_index_path is not None:
        logger.info(f"Loading index from: {opt.load_index_path} with index mode: {opt.index_mode}")
        index.load_index(opt.load_index_path, opt.save_index_n_shards)
        passages = [index.doc_map[i] for i in range(len(index.doc_map))]
    else:
        logger.info(f"Loading passages from: {opt.passages}")
        passages = []
        if not opt.use_file_passages:
            passages = load_passages(opt.passages, opt.max_passages)
            index.init_embeddings(passages)

    # optional, not a reference option: `opt.passage_store_path` = where the node-local passage store lives (e.g. under
    # /dev/shm). With it search_knn resolves the winners' text locally instead of exchanging it.
    store_path = getattr(opt, "passage_store_path", None)
    if store_path:
        if opt.load_index_path is not None:
            make = lambda: PassageStore.iter_saved_index(opt.load_index_path, opt.save_index_n_shards)   # noqa: E731
        else:
            make = lambda: PassageStore.iter_jsonl(opt.passages, opt.max_passages)                        # noqa: E731
        index.attach_passage_store(PassageStore.open_shared(store_path, make))

    return index, passages
