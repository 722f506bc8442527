"""Pins the oracle: every known-answer anchor of SURVEY.md Appendix C (stream size + FNV-1a64 minted from the
reference library, all modes 0-8 incl. RLE and STORED fallbacks) must be reproduced by the CPU restatement.

The reference ships no golden vectors of its own and oracle/_ref cannot be built under
this project's rules, so this table is what ties oracle/ to the reference.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ANCHORS = json.load(open(os.path.join(HERE, "golden", "anchors.json")))


def _id(a):
    return "cfg%s-%dx%dx%d-t%d-m%d-g%d%s" % (a["cfg"], a["w"], a["h"], a["bands"], a["dtype"], a["mode"], a["gen"],
                                              "-cb" if a["explicit_cb"] else "")


@pytest.mark.parametrize("a", ANCHORS, ids=_id)
def test_anchor(oracle, a):
    img = oracle.generate(a["w"], a["h"], a["bands"], a["dtype"], a["gen"], a["seed"])
    assert oracle.fnv(img) == a["fnv_in"], "generator does not match the anchor's input"
    cb = None
    if a["explicit_cb"]:
        cb = [1, 1, 1] + list(range(3, a["bands"]))
    stream = oracle.encode(img, a["dtype"], a["mode"], cband=cb)
    assert len(stream) == a["size"]
    assert oracle.fnv(stream) == a["fnv_stream"]
    assert stream[10] == a["hdr_mode"]
    out, dims, dt, mode = oracle.decode(stream)          # reference behaviour, defects included
    assert out is not None and dims == (a["w"], a["h"], a["bands"]) and dt == a["dtype"]
    same = np.array_equal(out, img.view(np.uint8).ravel())
    assert same == a["roundtrip"]
    if not a["roundtrip"]:
        # the reference itself does not round-trip here (defect B-1); its wrong output is anchored too,
        # and the spec-correct identity map does round-trip
        assert oracle.fnv(out) == a["ref_decoded_fnv"]
        out2, _, _, _ = oracle.decode(stream, identity=True)
        assert np.array_equal(out2, img.view(np.uint8).ravel())
