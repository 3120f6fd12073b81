"""oracle/model_oracle.py against the published stream grammar: the bytes of a tiny model.bin.gz spelled
out by hand from the Java Object Serialization Specification (6.4.2) and the DataOutput calls of
GenerationSerializer.writeObject (GS:96-105).  No reference-written file exists to pin against
("parity unpinned" in the oracle's header)."""
import gzip

import pytest

from oracle import model_oracle as mo

TINY = {"knownItemIDs": {5: [7]}, "X": {5: [1.0, -2.0]}, "Y": {7: [0.5, 0.25]},
        "itemTagIDs": [], "userTagIDs": [], "userClusters": [], "itemClusters": []}

TINY_STREAM = bytes.fromhex(
    "aced0005"                                   # STREAM_MAGIC, STREAM_VERSION
    "73" "72"                                    # TC_OBJECT, TC_CLASSDESC
    "0031") + b"net.myrrix.online.generation.GenerationSerializer" + bytes.fromhex(
    "0000000000000001"                           # serialVersionUID = 1L (GS:51)
    "03"                                         # SC_SERIALIZABLE | SC_WRITE_METHOD
    "0001" "4c" "000a") + b"generation" + bytes.fromhex(
    "74" "0029") + b"Lnet/myrrix/online/generation/Generation;" + bytes.fromhex(
    "78" "70"                                    # TC_ENDBLOCKDATA (class annotation), TC_NULL (no superclass)
    "77" "58"                                    # TC_BLOCKDATA, 88 bytes
    "00000001" "0000000000000005" "00000001" "0000000000000007"      # knownItemIDs {5: {7}}
    "00000001" "0000000000000005" "00000002" "3f800000" "c0000000"    # X {5: [1, -2]}
    "00000001" "0000000000000007" "00000002" "3f000000" "3e800000"    # Y {7: [.5, .25]}
    "00000000" "00000000" "00000000" "00000000"                       # tag sets, clusters
    "78")                                        # TC_ENDBLOCKDATA


def test_tiny_model_bytes():
    assert mo.stream(TINY) == TINY_STREAM
    assert mo.parse(TINY_STREAM) == TINY


def test_null_known_ids_and_clusters_round_trip(tmp_path):
    model = {"knownItemIDs": None, "X": {-3: [0.5] * 3, 2 ** 40: [1.5, 2.5, -1.0]}, "Y": {9: [1.0, 2.0, 3.0]},
             "itemTagIDs": [11, -12], "userTagIDs": [13],
             "userClusters": [([1, 2, 3], [0.25, 0.5, 0.75]), ([], [])], "itemClusters": [([9], [1.0, 2.0, 3.0])]}
    p = tmp_path / "model.bin.gz"
    mo.write_generation(model, p)
    assert mo.read_generation(p) == model
    assert mo.stream(model)[-1] == mo.TC_ENDBLOCKDATA and mo.payload(model)[:4] == b"\xff\xff\xff\xff"


@pytest.mark.parametrize("rows", [20, 21, 22, 300])
def test_records_are_1024_bytes_and_values_straddle_them(rows):
    # 4 (known = null) + 4 + rows * (8 + 4 + 4*9) + ... : record boundaries fall inside values
    model = {"knownItemIDs": None, "X": {i: [float(i + j) for j in range(9)] for i in range(rows)}, "Y": {}}
    s = mo.stream(model)
    head = len(mo.class_header())
    data = mo.payload(model)
    pos, sizes = head, []
    while s[pos] != mo.TC_ENDBLOCKDATA:
        if s[pos] == mo.TC_BLOCKDATA:
            n, pos = s[pos + 1], pos + 2
        else:
            assert s[pos] == mo.TC_BLOCKDATALONG
            n, pos = int.from_bytes(s[pos + 1:pos + 5], "big"), pos + 5
        sizes.append(n)
        pos += n
    assert pos == len(s) - 1 and sum(sizes) == len(data)
    assert all(n == 1024 for n in sizes[:-1]) and 0 < sizes[-1] <= 1024
    assert mo.parse(s)["X"] == model["X"]
    assert mo.parse(mo.stream(model, block=7))["X"] == model["X"]          # any record size reads back


def test_errors():
    with pytest.raises(ValueError):
        mo.payload({"knownItemIDs": None, "X": {1: [float("nan")]}, "Y": {}})
    with pytest.raises(ValueError):
        mo.write_generation(TINY, "/tmp/model.bin")
    with pytest.raises(IOError):
        mo.parse(TINY_STREAM[:-20] + b"\x78")
    bad = bytearray(TINY_STREAM)
    bad[4 + 2 + 2 + 49 + 7] = 2                                            # serialVersionUID 2
    with pytest.raises(IOError):
        mo.parse(bytes(bad))


def test_gzip_container(tmp_path):
    p = tmp_path / "m.bin.gz"
    mo.write_generation(TINY, p)
    raw = open(p, "rb").read()
    assert raw[:2] == b"\x1f\x8b" and gzip.decompress(raw) == TINY_STREAM
