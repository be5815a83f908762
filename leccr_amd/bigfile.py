"""Reader for the reference's BigFile feature store (LECCR/utils/bigfile.py:4-64): a directory with

    shape.txt    "<rows> <dims>"
    id.txt       whitespace-separated row names (ISO-8859-1)
    feature.bin  rows x dims float32, row-major

Same constructor and `read` / `read_one` / `shape` behaviour (requested names are de-duplicated, unknown names dropped,
the result is ordered by row number), served from one read-only memory map: a request is a single fancy-indexed
gather instead of a seek + fromfile per row, and `read_array` returns the rows as one float32 matrix ready for
`torch.from_numpy(...).pin_memory()` (video frame features are fed to the model as [b, frames, dims] batches).
"""
import os

import numpy as np


class BigFile:
    def __init__(self, datadir):
        with open(os.path.join(datadir, 'shape.txt')) as fh:
            self.nr_of_images, self.ndims = (int(x) for x in fh.readline().split())
        with open(os.path.join(datadir, 'id.txt'), 'rb') as fh:
            self.names = [str(n, encoding='ISO-8859-1') for n in fh.read().strip().split()]
        assert len(self.names) == self.nr_of_images
        self.name2index = dict(zip(self.names, range(self.nr_of_images)))
        self.binary_file = os.path.join(datadir, 'feature.bin')
        expect = self.nr_of_images * self.ndims * 4
        got = os.path.getsize(self.binary_file)
        if got != expect:
            raise ValueError(f'{self.binary_file}: {got} bytes, shape.txt says {self.nr_of_images} x {self.ndims} float32 = {expect}')
        self._map = np.memmap(self.binary_file, dtype=np.float32, mode='r', shape=(self.nr_of_images, self.ndims))
        print('[%s] %dx%d instances loaded from %s' % (self.__class__.__name__, self.nr_of_images, self.ndims, datadir))

    def _rows(self, requested, isname):
        requested = set(requested)
        if isname:
            rows = sorted(self.name2index[x] for x in requested if x in self.name2index)
        else:
            rows = sorted(int(x) for x in requested)
            if rows:
                assert rows[0] >= 0 and rows[-1] < len(self.names)
        return rows

    def read_array(self, requested, isname=True):
        """-> (names, float32 array [k, ndims]), rows in increasing row number."""
        rows = self._rows(requested, isname)
        if not rows:
            return [], np.zeros((0, self.ndims), dtype=np.float32)
        return [self.names[r] for r in rows], np.ascontiguousarray(self._map[np.asarray(rows)])

    def read(self, requested, isname=True):
        """The reference's return type: (names, list of per-row python lists)."""
        names, arr = self.read_array(requested, isname)
        if not names:
            return [], []
        return names, [row.tolist() for row in arr]

    def read_one(self, name):
        _, vectors = self.read([name])
        return vectors[0]

    def shape(self):
        return [self.nr_of_images, self.ndims]
