"""The reference's on-disk row format (src/tools/tsv/tsv_io.py): `<name>.tsv` with tab-separated columns, one row per
line; `<name>.lineidx` = decimal byte offset of every row, one per line; `<name>.lineidx.8b` = the same offsets as
little-endian uint64 (tsv_io.py:174-300 reader, 959-998 writer).  Image sets are rows of (key, base64 JPEG), caption
sets rows of (key, JSON list of {"caption": ...}), predictions rows of (key, JSON [{"caption", "conf"}])."""
import os
import os.path as op


class TSVFile(object):
    def __init__(self, tsv_file):
        self.tsv_file = tsv_file
        self.lineidx = op.splitext(tsv_file)[0] + '.lineidx'
        self.lineidx_8b = self.lineidx + '.8b'
        self._fp = None
        self._fp8b = None
        self._offsets = None
        self._pid = None

    # ------------------------------------------------------------------ index
    def _ensure_open(self):
        if self._fp is None or self._pid != os.getpid():        # re-open after a fork (DataLoader workers)
            self.close()
            self._fp = open(self.tsv_file, 'rb')
            self._pid = os.getpid()

    def close(self):
        for f in (self._fp, self._fp8b):
            if f is not None:
                f.close()
        self._fp = self._fp8b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_rows(self):
        if op.isfile(self.lineidx_8b):
            return op.getsize(self.lineidx_8b) // 8
        return len(self._load_offsets())

    __len__ = num_rows

    def _load_offsets(self):
        if self._offsets is None:
            if not op.isfile(self.lineidx):
                raise FileNotFoundError('%s has no .lineidx (write it with vitcap_amd.tsv.tsv_writer or generate_lineidx)' %
                                        self.tsv_file)
            with open(self.lineidx, 'r') as fp:
                self._offsets = tuple(int(l.strip()) for l in fp if l.strip())
        return self._offsets

    def get_offset(self, idx):
        if idx < 0 or idx >= self.num_rows():
            raise IndexError(idx)
        if op.isfile(self.lineidx_8b):
            if self._fp8b is None or self._pid != os.getpid():
                self._ensure_open()
                self._fp8b = open(self.lineidx_8b, 'rb')
            self._fp8b.seek(idx * 8)
            return int.from_bytes(self._fp8b.read(8), 'little')
        return self._load_offsets()[idx]

    # ------------------------------------------------------------------ rows
    def seek(self, idx):
        pos = self.get_offset(idx)
        self._ensure_open()
        self._fp.seek(pos)
        return [s.strip() for s in self._fp.readline().decode('utf-8').split('\t')]

    __getitem__ = seek

    def get_key(self, idx):
        return self.seek(idx)[0]

    def __iter__(self):
        for i in range(self.num_rows()):
            yield self.seek(i)


def tsv_writer(rows, tsv_file, sep='\t'):
    """Writes the .tsv, its .lineidx and .lineidx.8b; temporary files renamed at the end like the reference's writer."""
    d = op.dirname(tsv_file)
    if d:
        os.makedirs(d, exist_ok=True)
    lineidx = op.splitext(tsv_file)[0] + '.lineidx'
    names = (tsv_file, lineidx, lineidx + '.8b')
    sepb = sep.encode()
    off = 0
    with open(names[0] + '.tmp', 'wb') as fp, open(names[1] + '.tmp', 'w') as fi, open(names[2] + '.tmp', 'wb') as f8:
        for row in rows:
            line = sepb.join(v if isinstance(v, bytes) else str(v).encode('utf-8') for v in row) + b'\n'
            fp.write(line)
            fi.write('%d\n' % off)
            f8.write(off.to_bytes(8, 'little'))
            off += len(line)
    for n in names:
        os.replace(n + '.tmp', n)


def generate_lineidx(tsv_file):
    lineidx = op.splitext(tsv_file)[0] + '.lineidx'
    off = 0
    with open(tsv_file, 'rb') as fp, open(lineidx + '.tmp', 'w') as fi, open(lineidx + '.8b.tmp', 'wb') as f8:
        for line in fp:
            fi.write('%d\n' % off)
            f8.write(off.to_bytes(8, 'little'))
            off += len(line)
    os.replace(lineidx + '.tmp', lineidx)
    os.replace(lineidx + '.8b.tmp', lineidx + '.8b')
