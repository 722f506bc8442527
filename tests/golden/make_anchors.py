#!/usr/bin/env python3
"""Extract the known-answer anchor table of SURVEY.md Appendix C into tests/golden/anchors.json.

The anchors (input FNV-1a64, stream byte count, stream FNV-1a64, header mode, round-trip verdict and,
where the reference itself does not round-trip, the FNV of the reference's decoded output) were
minted by the surveyor from the reference library; this script only reformats that table so the
tests can iterate over it.  Run from the repo root:  python tests/golden/make_anchors.py
"""
import json, re, pathlib

ROOT = pathlib.Path(__file__).resolve().parents[2]
GEN = {"GRAD": 0, "NOISY3": 1, "LANDSAT16": 2, "DEM": 3, "TERRACE": 4, "FEW": 5, "PALETTE": 6, "RANDOM": 7, "RUNG63": 8}
DT = {"U8": 0, "I8": 1, "U16": 2, "I16": 3, "U32": 4, "I32": 5, "U64": 6, "I64": 7}

rows = []
for line in (ROOT / "SURVEY.md").read_text().splitlines():
    m = re.match(r"\|\s*(\w+)\s*\|\s*(\d+)×(\d+)×(\d+)\s*\|\s*(\w+)\s*\|\s*(\d+)→(\d+)\s*\|\s*(\w+) seed=(\d+)( cb=explicit)?\s*\|"
                 r"\s*([0-9a-f]{16})\s*\|\s*([\d  ]+)\|\s*([0-9a-f]{16})\s*\|\s*([^|]*)\|", line)
    if not m:
        continue
    cfg, w, h, b, dt, mode, hdr, gen, seed, cb, fin, size, fout, rt = m.groups()
    ref_out = re.search(r"ref output fnv ([0-9a-f]{16})", rt)
    rows.append(dict(cfg=cfg, w=int(w), h=int(h), bands=int(b), dtype=DT[dt], mode=int(mode), hdr_mode=int(hdr),
                     gen=GEN[gen], seed=int(seed), explicit_cb=bool(cb), fnv_in=fin,
                     size=int(re.sub(r"\D", "", size)), fnv_stream=fout,
                     roundtrip=rt.strip().startswith("=="), ref_decoded_fnv=ref_out.group(1) if ref_out else None))
(ROOT / "tests/golden/anchors.json").write_text(json.dumps(rows, indent=0) + "\n")
print(len(rows), "anchors")
