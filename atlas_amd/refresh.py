"""Streamed index refresh: the loop of `Atlas.build_index` (src/atlas.py:61-88) as a two-stream pipeline.

The reference tokenises a batch, moves it to the GPU, runs the fp16 retriever copy and scatters the embeddings into
`index.embeddings[:, a:b]`, one batch after the other. Here tokenised batches (host tensors) are staged through pinned
buffers and copied on a COPY stream while the encoder works on the previous batch on the compute stream; the pooled rows
are written straight into the slab (`Contriever.embed_into`), and the host only blocks when all staging slots are in
flight. Nothing is synchronised per batch.
"""
from typing import Iterable, Tuple

import torch


class IndexRefresher:
    def __init__(self, index, contriever_fp16, max_batch: int, max_len: int, depth: int = 3):
        """index: HipDistributedIndex with its slab allocated (init_embeddings); contriever_fp16: atlas_amd.retrievers.Contriever
        in fp16 on the slab's device (the `.half().eval()` copy of atlas.py:59)."""
        self.index, self.enc = index, contriever_fp16
        self.dev = index._slab.device
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.depth = depth
        mk = lambda: torch.empty((max_batch, max_len), dtype=torch.int64).pin_memory()     # noqa: E731
        self._pin = [(mk(), mk()) for _ in range(depth)]
        self._dev = [(torch.empty((max_batch, max_len), dtype=torch.int64, device=self.dev),
                      torch.empty((max_batch, max_len), dtype=torch.int64, device=self.dev)) for _ in range(depth)]
        self._ready = [torch.cuda.Event() for _ in range(depth)]      # H2D of the slot finished
        self._free = [torch.cuda.Event() for _ in range(depth)]       # encoder finished reading the slot
        self._used = [False] * depth

    @torch.no_grad()
    def run(self, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], row_offset: int = 0) -> int:
        """batches: (input_ids, attention_mask) host tensors [n, L] (n <= max_batch, L <= max_len), in slab row order
        starting at row_offset. Returns the number of rows written. Asynchronous: call torch.cuda.synchronize() (or use
        the slab on the current stream) afterwards."""
        compute = torch.cuda.current_stream(self.dev)
        row = row_offset
        for i, (ids, mask) in enumerate(batches):
            s = i % self.depth
            n, L = ids.shape
            if self._used[s]:
                self._free[s].synchronize()                            # only when `depth` batches are already in flight
            pi, pm = self._pin[s]
            pi[:n, :L].copy_(ids)
            pm[:n, :L].copy_(mask)
            di, dm = self._dev[s]
            with torch.cuda.stream(self.copy_stream):
                ids_d = di.view(-1)[: n * L].view(n, L)                # contiguous [n, L] views of the staging buffers
                mask_d = dm.view(-1)[: n * L].view(n, L)
                ids_d.copy_(pi[:n, :L], non_blocking=True)
                mask_d.copy_(pm[:n, :L], non_blocking=True)
                self._ready[s].record(self.copy_stream)
            compute.wait_event(self._ready[s])
            self.enc.embed_into(self.index._slab[row: row + n], ids_d, mask_d)
            self._free[s].record(compute)
            self._used[s] = True
            row += n
        self.index._pmax = None                                        # row norms changed: re-certify on the next search
        return row - row_offset
