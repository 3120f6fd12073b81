"""CPU restatement of the reference's model file (TEST INFRASTRUCTURE ONLY -- see oracle/README or
DESIGN.md section 2: only tests/ may import this).

Follows GenerationSerializer (online-local/src/net/myrrix/online/generation/GenerationSerializer.java):
writeObject :96-105, readObject :107-127, known ids :129-165, matrices :167-201, id sets :203-222,
clusters :224-262; container IOUtils.writeObjectToFile / readObjectFromFile
(common/src/net/myrrix/common/io/IOUtils.java:259-283): gzip around a Java Object Serialization
stream.  java.io.ObjectOutputStream is a JDK class, not part of the reference tree: its published
stream grammar (Java Object Serialization Specification, 6.4.2) is restated here with `struct`.

PARITY UNPINNED: the reference's tests hold no serialized model and there is no JVM in the image to
produce one, so this restatement is checked against the grammar only (tests/test_model_oracle.py
spells out the expected bytes of a tiny model by hand).

A model is a dict: knownItemIDs {user id: [item ids]} | None, X / Y {id: [floats]}, itemTagIDs /
userTagIDs [ids], userClusters / itemClusters [(members, centroid)].
"""
import gzip
import math
import struct

STREAM_MAGIC, STREAM_VERSION = 0xACED, 5
TC_NULL, TC_REFERENCE, TC_CLASSDESC, TC_OBJECT, TC_STRING = 0x70, 0x71, 0x72, 0x73, 0x74
TC_BLOCKDATA, TC_ENDBLOCKDATA, TC_BLOCKDATALONG, TC_LONGSTRING = 0x77, 0x78, 0x7A, 0x7C
SC_WRITE_METHOD, SC_SERIALIZABLE = 0x01, 0x02
CLASS_NAME = b"net.myrrix.online.generation.GenerationSerializer"
NULL_COUNT = -1  # GS:53


def _utf(b):
    return struct.pack(">H", len(b)) + b


def class_header():
    """TC_OBJECT + the class descriptor of GenerationSerializer: serialVersionUID 1 (GS:51), a private
    writeObject (SC_WRITE_METHOD), one serializable field `Generation generation` (GS:55)."""
    return (struct.pack(">HH", STREAM_MAGIC, STREAM_VERSION) + bytes([TC_OBJECT, TC_CLASSDESC]) + _utf(CLASS_NAME)
            + struct.pack(">q", 1) + bytes([SC_SERIALIZABLE | SC_WRITE_METHOD]) + struct.pack(">H", 1)
            + b"L" + _utf(b"generation") + bytes([TC_STRING]) + _utf(b"Lnet/myrrix/online/generation/Generation;")
            + bytes([TC_ENDBLOCKDATA, TC_NULL]))


def payload(model):
    """The DataOutput bytes of writeObject (GS:96-105), before they are cut into records."""
    out = []
    known = model.get("knownItemIDs")
    if known is None:
        out.append(struct.pack(">i", NULL_COUNT))
    else:
        out.append(struct.pack(">i", len(known)))
        for uid, items in known.items():
            out.append(struct.pack(">qi", uid, len(items)))
            out.append(struct.pack(">%dq" % len(items), *items))
    for name in ("X", "Y"):
        matrix = model.get(name) or {}
        out.append(struct.pack(">i", len(matrix)))
        for rid, row in matrix.items():
            if not all(math.isfinite(v) for v in row):
                raise ValueError("IllegalStateException: non-finite factor")     # GS:196
            out.append(struct.pack(">qi", rid, len(row)))
            out.append(struct.pack(">%df" % len(row), *row))
    for name in ("itemTagIDs", "userTagIDs"):
        ids = list(model.get(name) or [])
        out.append(struct.pack(">i%dq" % len(ids), len(ids), *ids))
    for name in ("userClusters", "itemClusters"):
        clusters = model.get(name) or []
        out.append(struct.pack(">i", len(clusters)))
        for members, centroid in clusters:
            out.append(struct.pack(">i%dq" % len(members), len(members), *members))
            out.append(struct.pack(">i%df" % len(centroid), len(centroid), *centroid))
    return b"".join(out)


def stream(model, block=1024):
    """The whole serialization stream; `block` is ObjectOutputStream's record size (1024 in the JDK,
    other values only to exercise readers)."""
    data = payload(model)
    out = [class_header()]
    for off in range(0, len(data), block):
        rec = data[off:off + block]
        out.append(bytes([TC_BLOCKDATA, len(rec)]) if len(rec) <= 0xFF else bytes([TC_BLOCKDATALONG]) + struct.pack(">i", len(rec)))
        out.append(rec)
    out.append(bytes([TC_ENDBLOCKDATA]))
    return b"".join(out)


def write_generation(model, path, block=1024):
    if not str(path).endswith(".gz"):
        raise ValueError("File should end in .gz: %s" % path)               # IOUtils.java:276
    with gzip.open(path, "wb") as f:
        f.write(stream(model, block))


class _Records:
    def __init__(self, data, pos):
        self.data, self.pos, self.left = data, pos, 0

    def take(self, n):
        out = b""
        while n:
            while not self.left:
                tc = self.data[self.pos]
                if tc == TC_BLOCKDATA:
                    self.left, self.pos = self.data[self.pos + 1], self.pos + 2
                elif tc == TC_BLOCKDATALONG:
                    self.left, self.pos = struct.unpack_from(">i", self.data, self.pos + 1)[0], self.pos + 5
                else:
                    raise IOError("EOFException / StreamCorruptedException: tag 0x%02x in the model data" % tc)
            m = min(n, self.left)
            if self.pos + m > len(self.data):
                raise IOError("EOFException: truncated stream")
            out += self.data[self.pos:self.pos + m]
            self.pos, self.left, n = self.pos + m, self.left - m, n - m
        return out

    def unpack(self, fmt):
        return struct.unpack(">" + fmt, self.take(struct.calcsize(">" + fmt)))


def parse(data):
    head = class_header()
    if data[:len(head)] != head:
        raise IOError("StreamCorruptedException / InvalidClassException: unexpected stream or class header")
    r = _Records(data, len(head))
    model = {}
    n, = r.unpack("i")
    if n == NULL_COUNT:
        model["knownItemIDs"] = None
    else:
        known = {}
        for _ in range(n):
            uid, cnt = r.unpack("qi")
            known[uid] = list(r.unpack("%dq" % cnt))
        model["knownItemIDs"] = known
    for name in ("X", "Y"):
        matrix = {}
        for _ in range(r.unpack("i")[0]):
            rid, k = r.unpack("qi")
            row = list(r.unpack("%df" % k))
            if not all(math.isfinite(v) for v in row):
                raise ValueError("IllegalStateException: non-finite factor")     # GS:175
            matrix[rid] = row
        model[name] = matrix
    for name in ("itemTagIDs", "userTagIDs"):
        model[name] = list(r.unpack("%dq" % r.unpack("i")[0]))
    for name in ("userClusters", "itemClusters"):
        clusters = []
        for _ in range(r.unpack("i")[0]):
            members = list(r.unpack("%dq" % r.unpack("i")[0]))
            centroid = list(r.unpack("%df" % r.unpack("i")[0]))
            clusters.append((members, centroid))
        model[name] = clusters
    return model


def read_generation(path):
    with gzip.open(path, "rb") as f:
        return parse(f.read())
