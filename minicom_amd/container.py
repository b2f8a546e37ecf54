"""The `.minicom` container: what the reference's `minicom` script does with the stream files after the binary has
written them (reference minicom:110-172 single end, :232-293 paired end, :304-403 the way back).

The reference groups the per-thread streams into inner tars (`refbin.tar`, `dirbin.tar`, `begposbin.tar`,
`dif_char.tar`, with `-p` `idsbin.tar`, paired end `peidsbin.tar` and `filebin.tar`), runs every inner tar and the
five single files through the external `bsc` (7z for `peidsbin.tar`) and tars the results together with `info.txt`.
Those binaries are fetched over the network by the reference's `install.sh` and do not exist here, so the entropy
stage is built in: every member goes through one of the codecs below and carries its codec as file extension.
`codec="bsc"` runs a `bsc` binary found on PATH with the reference's flags and then produces the reference's own
member names; everything else about the layout is the reference's, so an archive written with `bsc` present is
what the reference's script unpacks.

This is packaging on the host, nothing here touches the GPU; the stream files themselves are the parity boundary
(tests/test_streams.py compares them byte for byte with the reference's).
"""
import bz2
import glob
import io
import lzma
import os
import shutil
import subprocess
import tarfile
import zlib
from concurrent.futures import ThreadPoolExecutor

# inner tar -> the stream files it takes (shell patterns of the reference script)
GROUPS = (
    ("idsbin", ("ids.bin.*", "*.ids.bin")),          # minicom:110-118, only with -p
    ("dif_char", ("dif_char.txt.*",)),               # :121-124
    ("begposbin", ("beg_pos.bin.*",)),               # :126-129
    ("peidsbin", ("peids.bin.*",)),                  # :243-247, paired end only
    ("refbin", ("ref.bin.*",)),                      # :131-134
    ("dirbin", ("dir.bin.*",)),                      # :139-142
    ("filebin", ("file.bin.*",)),                    # :259-262, paired end only
)
SINGLES = ("single_N.seq", "single.seq", "AA.txt", "TT.txt", "NN.txt")      # :136-137, :144-146
CODECS = ("xz", "bz2", "gz", "raw", "bsc")


def _encode(data: bytes, codec: str) -> bytes:
    if codec == "xz":
        return lzma.compress(data, preset=6)
    if codec == "bz2":
        return bz2.compress(data, 9)
    if codec == "gz":
        return zlib.compress(data, 6)
    if codec == "raw":
        return data
    raise ValueError("unknown codec " + codec)


def _decode(data: bytes, codec: str) -> bytes:
    if codec == "xz":
        return lzma.decompress(data)
    if codec == "bz2":
        return bz2.decompress(data)
    if codec == "gz":
        return zlib.decompress(data)
    if codec == "raw":
        return data
    raise ValueError("unknown codec " + codec)


def _bsc(args, src: bytes, tmpdir: str, tag: str) -> bytes:
    exe = shutil.which("bsc")
    if not exe:
        raise RuntimeError("codec 'bsc' needs a bsc binary on PATH (the reference's install.sh fetches one)")
    a, b = os.path.join(tmpdir, tag + ".in"), os.path.join(tmpdir, tag + ".out")
    with open(a, "wb") as f:
        f.write(src)
    subprocess.run([exe, args[0], a, b] + list(args[1:]), check=True, stdout=subprocess.DEVNULL)
    with open(b, "rb") as f:
        out = f.read()
    os.remove(a); os.remove(b)
    return out


def _inner_tar(folder: str, names) -> bytes:
    """`tar -cf X.tar -C X .` of the reference: plain member names, sorted for a reproducible archive."""
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w", format=tarfile.GNU_FORMAT) as t:
        for n in sorted(names):
            ti = tarfile.TarInfo(n)
            ti.size = os.path.getsize(os.path.join(folder, n))
            ti.mode = 0o644
            with open(os.path.join(folder, n), "rb") as f:
                t.addfile(ti, f)
    return buf.getvalue()


def pack(folder: str, out_path: str, codec: str = "xz", threads: int = 8) -> dict:
    """Stream files in `folder` (as cluster_dump left them) -> one `.minicom` file.  Returns the member sizes."""
    if codec not in CODECS:
        raise ValueError("codec must be one of %s" % (CODECS,))
    if not os.path.isfile(os.path.join(folder, "info.txt")):
        raise FileNotFoundError("no info.txt in %s: not a stream directory" % folder)
    members = []                                         # (name without codec extension, bytes)
    for group, patterns in GROUPS:
        names = sorted({os.path.basename(p) for pat in patterns for p in glob.glob(os.path.join(folder, pat))})
        if names:
            members.append((group + ".tar", _inner_tar(folder, names)))
    for s in SINGLES:
        p = os.path.join(folder, s)
        if os.path.isfile(p):
            with open(p, "rb") as f:
                members.append((s, f.read()))
    ext = codec

    def enc(item):
        name, data = item
        if codec == "bsc":                               # minicom:115 etc.: bsc e IN OUT -b64p -tN -e2
            return name + ".bsc", _bsc(("e", "-b64p", "-t%d" % threads, "-e2"), data, folder, name)
        return name + "." + ext, _encode(data, codec)

    with ThreadPoolExecutor(max(1, threads)) as ex:
        packed = list(ex.map(enc, members))
    sizes = {}
    with tarfile.open(out_path, mode="w", format=tarfile.GNU_FORMAT) as t:
        with open(os.path.join(folder, "info.txt"), "rb") as f:
            info = f.read()
        for name, data in [("info.txt", info)] + packed:
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            ti.mode = 0o644
            t.addfile(ti, io.BytesIO(data))
            sizes[name] = len(data)
    return sizes


def unpack(path: str, folder: str, threads: int = 8) -> dict:
    """`.minicom` file -> the stream files in `folder` (created if absent).  Returns what the archive says about itself:
    {"order": bool, "paired": bool} as the reference's script decides them (minicom:326-334)."""
    os.makedirs(folder, exist_ok=True)
    with tarfile.open(path, mode="r") as t:
        items = [(m.name.lstrip("./"), t.extractfile(m).read()) for m in t.getmembers() if m.isfile()]
    kinds = {"order": False, "paired": False}

    def dec(item):
        name, data = item
        if name == "info.txt":
            return name, data
        base, ext = name.rsplit(".", 1)
        if ext == "bsc":
            return base, _bsc(("d", "-t%d" % threads), data, folder, base)
        if ext == "7z":
            raise RuntimeError("member %s needs 7z; archives written here use one codec for every member" % name)
        return base, _decode(data, ext)

    with ThreadPoolExecutor(max(1, threads)) as ex:
        plain = list(ex.map(dec, items))
    for name, data in plain:
        if name.startswith("idsbin.tar"):
            kinds["order"] = True
        if name.startswith("filebin.tar"):
            kinds["paired"] = True
        if name.endswith(".tar"):
            with tarfile.open(fileobj=io.BytesIO(data), mode="r") as t:
                for m in t.getmembers():
                    if m.isfile():
                        n = os.path.basename(m.name)
                        with open(os.path.join(folder, n), "wb") as f:
                            f.write(t.extractfile(m).read())
        else:
            with open(os.path.join(folder, name), "wb") as f:
                f.write(data)
    return kinds


# ---- end to end: what `minicom -r IN [-p]`, `minicom -1 IN1 -2 IN2` and `minicom -d X.minicom` amount to -------------
def compress_fastq(path: str, out_path: str, path2: str | None = None, order: bool = False, codec: str = "xz",
                   device: int = 0, threads: int = 8, **params) -> dict:
    """FASTQ/FASTA (plain or .gz; path2 = the mates' file) -> `.minicom`.  The hot path runs on `device` (there is no CPU
    fallback), the stream writer and the packaging on the host.  Returns pack()'s member sizes plus the read count."""
    import tempfile
    from .pipeline import Pipeline
    if order and path2 is not None:
        raise ValueError("-p is a single-end option (reference minicom:439-476)")
    p = Pipeline.from_fastq(path, device=device, path2=path2, host_threads=threads, **params)
    try:
        p.pre_process()
        with tempfile.TemporaryDirectory(dir=os.path.dirname(os.path.abspath(out_path)) or ".") as td:
            p.cluster_dump(td, order=order, paired=path2 is not None)
            sizes = pack(td, out_path, codec=codec, threads=threads)
        sizes["n_reads"] = p.n
        return sizes
    finally:
        p.close()


def decompress_file(path: str, out_path: str, out_path2: str | None = None, threads: int = 8) -> int:
    """`.minicom` -> reads, one per line: the original order for an archive written with -p, two files (line i of both a
    pair) for a paired-end archive.  Host only.  Returns the number of reads (pairs for paired end)."""
    import tempfile
    from .pipeline import decompress, decompress_pe
    with tempfile.TemporaryDirectory(dir=os.path.dirname(os.path.abspath(out_path)) or ".") as td:
        kinds = unpack(path, td, threads=threads)
        if kinds["paired"]:
            if out_path2 is None:
                raise ValueError("a paired-end archive decodes into two files")
            return decompress_pe(td, out_path, out_path2)
        return decompress(td, out_path, order=kinds["order"])
