#!/usr/bin/env python3
"""Reads the variable inventory (name -> shape) out of a TensorFlow checkpoint `.index` file without
TensorFlow: the file is a LevelDB-format table whose values are BundleEntryProto messages.  Used once,
in the build container, to turn /root/reference/bestrecord/model-229999.index into the data fixture
tests/golden/rfnet_variables.json (names and shapes only) that pins the parameter inventory of
rfnet_amd/rfnet.py.   usage: python tools/read_tf_index.py <index file> [out.json]"""
import json
import struct
import sys


def varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def read_block(data, offset, size):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed block (snappy) not supported")
    nrestart = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = varint(block, pos)
        non_shared, pos = varint(block, pos)
        vlen, pos = varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def parse_entry(buf):
    """BundleEntryProto: 1 dtype, 2 shape{2 dim{1 size}}, 3 shard, 4 offset, 5 size, 6 crc."""
    pos, dtype, shape, size = 0, None, [], None
    while pos < len(buf):
        tag, pos = varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = varint(buf, pos)
            if field == 1:
                dtype = v
            elif field == 5:
                size = v
        elif wt == 2:
            ln, pos = varint(buf, pos)
            sub = buf[pos:pos + ln]
            pos += ln
            if field == 2:  # TensorShapeProto
                sp = 0
                while sp < len(sub):
                    t2, sp = varint(sub, sp)
                    if t2 & 7 == 2:
                        l2, sp = varint(sub, sp)
                        dim = sub[sp:sp + l2]
                        sp += l2
                        dp = 0
                        while dp < len(dim):
                            t3, dp = varint(dim, dp)
                            if t3 & 7 == 0:
                                v3, dp = varint(dim, dp)
                                if t3 >> 3 == 1:
                                    shape.append(v3)
                            elif t3 & 7 == 2:
                                l3, dp = varint(dim, dp)
                                dp += l3
                    elif t2 & 7 == 0:
                        _, sp = varint(sub, sp)
        elif wt == 5:
            pos += 4
        elif wt == 1:
            pos += 8
    return dtype, shape, size


def read_index(path):
    data = open(path, "rb").read()
    footer = data[-48:]
    pos = 0
    _, pos = varint(footer, pos)
    _, pos = varint(footer, pos)
    ioff, pos = varint(footer, pos)
    isize, pos = varint(footer, pos)
    out = {}
    for _, handle in read_block(data, ioff, isize):
        boff, p2 = varint(handle, 0)
        bsize, _ = varint(handle, p2)
        for key, val in read_block(data, boff, bsize):
            name = key.decode("utf-8", "replace")
            if not name:
                continue  # header entry
            dtype, shape, size = parse_entry(val)
            out[name] = {"shape": shape, "bytes": size, "dtype": dtype}
    return out


if __name__ == "__main__":
    entries = read_index(sys.argv[1])
    model = {k: v["shape"] for k, v in sorted(entries.items())
             if "Adam" not in k and not k.endswith("_power") and k != "Variable"}
    print(len(entries), "entries,", len(model), "model variables,",
          sum(int(__import__("numpy").prod(s)) if s else 1 for s in model.values()), "parameters")
    if len(sys.argv) > 2:
        json.dump(model, open(sys.argv[2], "w"), indent=0, sort_keys=True)
    else:
        for k, v in model.items():
            print(k, v)
