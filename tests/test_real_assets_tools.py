"""The real-asset pipeline (scripts/real_assets.sh) is not dead code: on a STAND-IN asset directory (tools/make_fake_assets.py: random-init weights
in the HF layout, a made-up WordPiece vocabulary, a made-up corpus) the golden generator runs the reference's own modules on the checkpoint
directory with the real HF tokenizer call of src/atlas.py:66-75, and the fixture it writes is what the CPU restatement of the encoder computes
from the same directory. CPU part only (needs /root/reference); the GPU steps are dry-run by `scripts/gpu_session.sh <tag> realdry`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/root/reference/src/retrievers.py"), reason="needs the reference checkout")
def test_golden_from_a_checkpoint_directory_through_the_reference_modules(tmp_path):
    fake = str(tmp_path / "fake")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_fake_assets.py"), fake, "--layers", "2", "--passages", "120", "--queries", "8"],
                   check=True, capture_output=True, timeout=600)
    out = str(tmp_path / "enc_real.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden_real.py"), "--checkpoint", os.path.join(fake, "contriever"),
                        "--passages", os.path.join(fake, "passages.jsonl"), "--n", "10", "--max-length", "64", "--out", out],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    z = np.load(out)
    meta = json.loads(bytes(z["meta"]).decode())
    assert "real tokenizer" in meta["source"] and meta["layers"] == 2
    ids, mask = z["input_ids"], z["attention_mask"]
    assert ids.shape == mask.shape == (10, 64) or ids.shape[1] <= 64
    assert (ids[:, 0] == 101).all() and all(ids[i, mask[i].sum() - 1] == 102 for i in range(ids.shape[0]))     # [CLS] ... [SEP] from the vocab file
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, ROOT)
    from make_golden_real import checkpoint_sha

    from atlas_amd import retrievers
    from oracle.contriever_ref import BertConfigLite, ContrieverRef

    mine = retrievers.Contriever.from_pretrained(os.path.join(fake, "contriever"))
    assert checkpoint_sha(mine.state_dict()) == bytes(z["checkpoint_sha"]).decode()
    ref = ContrieverRef(BertConfigLite(num_hidden_layers=2))
    ref.load_state_dict(mine.state_dict(), strict=True)
    with torch.no_grad():
        e = ref.eval()(torch.from_numpy(ids), torch.from_numpy(mask)).numpy()
    assert np.abs(e - z["emb_fp32"]).max() <= 2e-6 * np.abs(e).max()
    # the queries of the stand-in set have their answers in the corpus (what tools/retrieve_only.py's recall check looks for)
    corpus = open(os.path.join(fake, "passages.jsonl")).read()
    for ln in open(os.path.join(fake, "queries.jsonl")):
        r = json.loads(ln)
        assert r["answers"][0] in corpus and r["question"]
