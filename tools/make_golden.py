#!/usr/bin/env python3
"""Pack the consensus-spec KZG vectors the reference's tests hold (/root/reference/tests, 368
data.yaml cases, ~90 MB of hex text) into compact committed fixtures under tests/golden/.

The vectors are DATA (inputs and expected outputs); no reference source is copied.  Large byte
strings (blobs, cells) are shared by many cases, so every value of 2048 bytes or more is stored
once in a content-addressed object file:

  tests/golden/objects.bin   concatenated unique byte strings
  tests/golden/objects.json  sha256-prefix -> [offset, length]
  tests/golden/cases.json    {function: {case: {"input": ..., "output": ...}}}; big values are
                             {"$obj": key}; small hex strings stay inline as "0x..".
  c-kzg-4844_amd/data/trusted_setup.txt  the mainnet setup (runtime data, sha256 d39b9f2d...26b7)

Run in the build container only (reads /root/reference):  python tools/make_golden.py
"""
import hashlib
import json
import os
import shutil
import sys

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
BIG = 2048

objects = {}
blob = bytearray()


def pack(v):
    if isinstance(v, str) and v.startswith("0x"):
        try:
            raw = bytes.fromhex(v[2:])
        except ValueError:
            return v  # deliberately malformed hex stays as text
        if len(raw) >= BIG:
            key = hashlib.sha256(raw).hexdigest()[:20]
            if key not in objects:
                objects[key] = [len(blob), len(raw)]
                blob.extend(raw)
            return {"$obj": key}
        return v
    if isinstance(v, list):
        return [pack(x) for x in v]
    if isinstance(v, dict):
        return {k: pack(x) for k, x in v.items()}
    return v


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    troot = os.path.join(REF, "tests")
    total = 0
    for fn in sorted(os.listdir(troot)):
        d = os.path.join(troot, fn, "kzg-mainnet")
        if not os.path.isdir(d):
            continue
        cases[fn] = {}
        for case in sorted(os.listdir(d)):
            with open(os.path.join(d, case, "data.yaml")) as f:
                y = yaml.load(f, Loader=yaml.CSafeLoader)
            cases[fn][case] = {"input": pack(y["input"]), "output": pack(y["output"])}
            total += 1
    with open(os.path.join(OUT, "objects.bin"), "wb") as f:
        f.write(blob)
    with open(os.path.join(OUT, "objects.json"), "w") as f:
        json.dump(objects, f)
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    shutil.copyfile(os.path.join(REF, "src", "trusted_setup.txt"),
                    os.path.join(OUT, "..", "..", "c-kzg-4844_amd", "data", "trusted_setup.txt"))
    print("cases:", total, "objects:", len(objects), "bytes:", len(blob))


if __name__ == "__main__":
    sys.exit(main())
