"""Test helper: a minimal BAM (BGZF) writer, so the native reader can be checked against SAM text
without samtools.  SAM spec v1 sections 4.1 (BGZF) and 4.2 (BAM records)."""
import struct
import zlib

OPS = "MIDNSHP=X"


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def parse_cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | OPS.index(ch))
            num = ""
    return out


def sam_to_bam(sam_text, bam_path, block=20000):
    refs, recs, header = [], [], []
    for line in sam_text.splitlines():
        if line.startswith("@"):
            header.append(line)
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
            continue
        if line.strip():
            recs.append(line.split("\t"))
    rid = {n: i for i, (n, _) in enumerate(refs)}
    text = ("\n".join(header) + "\n").encode()
    raw = bytearray(b"BAM\x01") + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, ln in refs:
        raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    for f in recs:
        name, flag, rname, pos = f[0].encode() + b"\0", int(f[1]), f[2], int(f[3]) - 1
        cig = [] if f[5] == "*" else parse_cigar(f[5])
        seq = "" if f[9] == "*" else f[9]
        reflen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        codes = "=ACMGRSVTWYHKDBN"
        sq = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            sq[i // 2] |= codes.index(ch.upper() if ch.upper() in codes else "N") << (4 if i % 2 == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", rid.get(rname, -1), pos, len(name), int(f[4]),
                           _reg2bin(pos, pos + max(reflen, 1)), len(cig), flag, len(seq),
                           rid.get(f[6] if f[6] != "=" else rname, -1), int(f[7]) - 1, int(f[8]))
        body += name + b"".join(struct.pack("<I", c) for c in cig) + bytes(sq) + b"\xff" * len(seq)
        raw += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as out:
        for i in range(0, len(raw), block):
            out.write(bgzf_block(bytes(raw[i:i + block])))
        out.write(bgzf_block(b""))   # EOF marker block
