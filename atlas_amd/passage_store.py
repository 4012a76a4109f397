"""Node-local passage store: the text of ALL passages, memory-mapped once per node, keyed by global passage id.

Why: `DistributedIndex.search_knn` returns passage dicts, and the winners of a query live on other
ranks. The reference moves W*k pickled passages per query through `varsize_gather` (src/index.py:134-150, ~160 ms of
host time per call); `HipDistributedIndex` without a store moves the k winners per query with one `all_gather_object`.
With a store attached there is NO text collective: after the packed (score, id) all-gather every rank resolves the
winners' ids locally. One copy per node (page cache / /dev/shm), shared by its ranks through mmap.

File format (two files, little-endian):
    <path>.off   int64[N + 1]  byte offsets into the blob; offsets[i] == offsets[i+1] means "None" (blank line, index_io.py:57-59)
    <path>.bin   concatenated UTF-8 JSON documents, one per passage, in GLOBAL ID order
Global id = what `HipDistributedIndex` puts in the packed candidates: the passage's line number over the jsonl files for
round-robin shards (src/index_io.py:41), the position in the concatenation of the saved shards for a loaded index.
"""
import json
import os
import pickle
from typing import Iterable, Optional

import numpy as np

from . import dist_utils


class PassageStore:
    def __init__(self, path: str):
        self.path = path
        self._off = np.load(path + ".off.npy", mmap_mode="r")
        self._bin = np.memmap(path + ".bin", dtype=np.uint8, mode="r") if os.path.getsize(path + ".bin") > 0 else np.zeros(0, np.uint8)
        assert self._off.ndim == 1 and self._off.shape[0] >= 1 and int(self._off[-1]) == self._bin.shape[0], "corrupt passage store"

    def __len__(self) -> int:
        return int(self._off.shape[0]) - 1

    def get(self, gid: int) -> Optional[dict]:
        a, b = int(self._off[gid]), int(self._off[gid + 1])
        if a == b:
            return None
        return json.loads(bytes(self._bin[a:b]).decode("utf-8"))

    __getitem__ = get

    # ------------------------------------------------------------------ builders
    @staticmethod
    def build_from_items(path: str, items: Iterable[Optional[dict]]) -> None:
        """items in global-id order (dicts, or None for blank lines). Written to temporaries and renamed: readers never
        see a partial store."""
        offs = [0]
        tmp_bin = path + ".bin.tmp%d" % os.getpid()
        with open(tmp_bin, "wb") as fb:
            for it in items:
                if it is not None:
                    fb.write(json.dumps(it, ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
                offs.append(fb.tell())
        tmp_off = path + ".off.tmp%d.npy" % os.getpid()
        np.save(tmp_off, np.asarray(offs, dtype=np.int64))
        os.replace(tmp_bin, path + ".bin")
        os.replace(tmp_off, path + ".off.npy")

    @staticmethod
    def iter_jsonl(filenames, maxload: int = -1):
        """every line of the passage files, parsed exactly like index_io.load_passages (title/section join, None for blank
        lines) but for ALL ranks: item c is global passage c"""
        counter = 0
        for filename in filenames:
            with open(filename) as fobj:
                for line in fobj:
                    if maxload > -1 and counter >= maxload:
                        return
                    if line.strip() != "":
                        item = json.loads(line)
                        assert "id" in item
                        if "title" in item and "section" in item and len(item["section"]) > 0:
                            item["title"] = f"{item['title']}: {item['section']}"
                        yield item
                    else:
                        yield None
                    counter += 1

    @staticmethod
    def iter_saved_index(index_dir: str, total_saved_shards: int):
        """the passages of a saved index (passages.{shard}.pt pickles, reference format) in shard order = the global ids of
        an index loaded with `load_index`"""
        for shard_id in range(total_saved_shards):
            with open(os.path.join(index_dir, f"passages.{shard_id}.pt"), "rb") as fobj:
                for p in pickle.load(fobj):
                    yield p

    @classmethod
    def open_shared(cls, path: str, make_items) -> "PassageStore":
        """Collective. The first rank of each node (LOCAL_RANK 0, or rank 0 without a launcher) builds the store from
        `make_items()` if it does not exist yet; everyone maps it after a barrier."""
        local_rank = int(os.environ.get("LOCAL_RANK", dist_utils.get_rank()))
        if local_rank == 0 and not (os.path.exists(path + ".off.npy") and os.path.exists(path + ".bin")):
            cls.build_from_items(path, make_items())
        if dist_utils.is_initialized():
            dist_utils.barrier()
        return cls(path)
