"""Binary PLY point-cloud files, the on-disk format either side of the hot path (first
half): same call signatures, return values and file bytes as the reference's ``utils/ply.py`` (``read_ply`` :114-194,
``write_ply`` :215-326), so that real DALES / Vaihingen3D tiles can feed ``weasal_amd.pyramid`` and predictions can
be written back.  Host-side numpy only; nothing here touches the GPU.

File layout (what both functions agree on): a text header ``ply / format binary_<endian>_endian 1.0 /
element vertex N / property <type> <name> ... / [element face M / property list uchar int vertex_indices] /
end_header`` followed by N packed vertex records and, for meshes, M records of (uchar 3, int32 x 3).
Pinned by tests/test_ply_cpu.py against files written by the reference's own writer (tests/golden/ply/)."""
import sys

import numpy as np

# PLY scalar type names (both spellings) -> numpy type codes
_PLY_TO_NUMPY = {
    'int8': 'i1', 'char': 'i1', 'uint8': 'u1', 'uchar': 'u1',
    'int16': 'i2', 'short': 'i2', 'uint16': 'u2', 'ushort': 'u2',
    'int32': 'i4', 'int': 'i4', 'uint32': 'u4', 'uint': 'u4',
    'float32': 'f4', 'float': 'f4', 'float64': 'f8', 'double': 'f8',
}
_BYTE_ORDER = {'binary_big_endian': '>', 'binary_little_endian': '<'}


def _read_header(fh, want_faces):
    """-> (n_vertices, n_faces or None, [(name, dtype str)]) ; leaves `fh` at the first data byte"""
    first = fh.readline()
    if b'ply' not in first:
        raise ValueError('The file does not start whith the word ply')
    fmt = fh.readline().split()[1].decode()
    if fmt == 'ascii':
        raise ValueError('The file is not binary')
    order = _BYTE_ORDER[fmt]
    n_vertices = n_faces = None
    props = []
    element = None
    while True:
        line = fh.readline()
        if line == b'' or b'end_header' in line:
            break
        tok = line.split()
        if not tok:
            continue
        if tok[0] == b'element':
            element = tok[1].decode()
            if want_faces and element == 'face':
                n_faces = int(tok[2])
            elif not want_faces or element == 'vertex':
                n_vertices = int(tok[2])      # point-cloud mode: the count of the last element line, like the reference
        elif tok[0] == b'property':
            if want_faces and element != 'vertex':
                continue                      # "property list uchar int vertex_indices": fixed record below
            props.append((tok[2].decode(), order + _PLY_TO_NUMPY[tok[1].decode()]))
    return n_vertices, n_faces, props, order


def read_ply(filename, triangular_mesh=False):
    """Structured array with one field per property (point cloud), or ``[vertices, faces[M,3] int32]`` for
    ``triangular_mesh=True``.  ASCII files are rejected like in the reference."""
    with open(filename, 'rb') as fh:
        n_vertices, n_faces, props, order = _read_header(fh, triangular_mesh)
        vertices = np.fromfile(fh, dtype=props, count=n_vertices)
        if not triangular_mesh:
            return vertices
        face_t = [('k', order + 'u1'), ('v1', order + 'i4'), ('v2', order + 'i4'), ('v3', order + 'i4')]
        faces = np.fromfile(fh, dtype=face_t, count=n_faces)
        return [vertices, np.vstack((faces['v1'], faces['v2'], faces['v3'])).T]


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """Writes the columns of `field_list` (an array, or a list / tuple of 1-D and 2-D arrays with the same number
    of rows) as vertex properties named `field_names`; optional `triangular_faces` [M,3].  Returns True, or False
    (after printing the reference's message) when the fields are inconsistent.  '.ply' is appended if missing."""
    fields = list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]
    columns = []
    for f in fields:
        if f.ndim > 2:
            print('fields have more than 2 dimensions')
            return False
        columns.append(f.reshape(-1, 1) if f.ndim < 2 else f)
    rows = [c.shape[0] for c in columns]
    if any(r != rows[0] for r in rows):
        print('wrong field dimensions')
        return False
    if sum(c.shape[1] for c in columns) != len(field_names):
        print('wrong number of field names')
        return False
    if not filename.endswith('.ply'):
        filename += '.ply'

    record, header = [], ['ply', 'format binary_%s_endian 1.0' % sys.byteorder, 'element vertex %d' % rows[0]]
    flat = []
    for c in columns:
        for col in c.T:
            name = field_names[len(flat)]
            header.append('property %s %s' % (col.dtype.name, name))
            record.append((name, col.dtype.str))
            flat.append(col)
    if triangular_faces is not None:
        header.append('element face %d' % triangular_faces.shape[0])
        header.append('property list uchar int vertex_indices')
    header.append('end_header')

    packed = np.empty(rows[0], dtype=record)
    for (name, _), col in zip(record, flat):
        packed[name] = col
    with open(filename, 'wb') as fh:
        fh.write(('\n'.join(header) + '\n').encode())
        packed.tofile(fh)
        if triangular_faces is not None:
            tri = triangular_faces.astype(np.int32)
            rec = np.empty(tri.shape[0], dtype=[('k', 'uint8'), ('0', 'int32'), ('1', 'int32'), ('2', 'int32')])
            rec['k'] = 3
            rec['0'], rec['1'], rec['2'] = tri[:, 0], tri[:, 1], tri[:, 2]
            rec.tofile(fh)
    return True
