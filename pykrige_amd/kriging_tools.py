"""Grid file I/O either side of execute(): ARC ASCII grids (*.asc) and ZMAP+ grids (*.zmap), same signatures, file
formats and return tuples as the reference (kriging_tools.py:23-127 write_asc_grid, 130-250 read_asc_grid, 253-352
write_zmap_grid, 355-459 read_zmap_grid).  Host text I/O -- nothing here touches the device."""
import datetime
import io
import os
import warnings

import numpy as np


def _plain_grid(x, y, z, no_data):
    """Fill a masked z with no_data, squeeze the three arrays, check the axes are regular; -> x, y, z, dx, dy."""
    if np.ma.is_masked(z):
        z = np.ma.filled(np.ma.asarray(z, dtype=float), no_data)
    x, y, z = np.squeeze(np.array(x)), np.squeeze(np.array(y)), np.squeeze(np.array(z))
    return x, y, z


def _spacing(x, y):
    dx, dy = abs(x[1] - x[0]), abs(y[1] - y[0])
    mean_dx, mean_dy = abs((x[-1] - x[0]) / (x.shape[0] - 1)), abs((y[-1] - y[0]) / (y.shape[0] - 1))
    if not np.isclose(mean_dx, dx) or not np.isclose(mean_dy, dy):
        raise ValueError("X or Y spacing is not constant; *.asc grid cannot be written.")
    return dx, dy


def write_asc_grid(x, y, z, filename="output.asc", no_data=-999.0, style=1):
    """Write z (M, N), given at the cell centres x (N,), y (M,), as an ARC ASCII grid.  style=1: DX/DY with
    XLLCENTER/YLLCENTER; style=2: CELLSIZE (needs dx == dy) with XLLCORNER/YLLCORNER."""
    x, y, z = _plain_grid(x, y, z, no_data)
    if z.ndim != 2:
        raise ValueError("Two-dimensional grid is required to write *.asc grid.")
    if x.ndim > 1 or y.ndim > 1:
        raise ValueError("Dimensions of X and/or Y coordinate arrays are not as expected. Could not write *.asc grid.")
    if z.shape != (y.size, x.size):
        warnings.warn("Grid dimensions are not as expected. Incorrect *.asc file generation may result.", RuntimeWarning)
    if np.amin(x) != x[0] or np.amin(y) != y[0]:
        warnings.warn("Order of X or Y coordinates is not as expected. Incorrect *.asc file generation may result.",
                      RuntimeWarning)
    dx, dy = _spacing(x, y)
    nrows, ncols = z.shape
    if style == 1:
        head = [("NCOLS", ncols), ("NROWS", nrows), ("XLLCENTER", x[0]), ("YLLCENTER", y[0]), ("DX", dx), ("DY", dy),
                ("NODATA_VALUE", no_data)]
    elif style == 2:
        if dx != dy:
            raise ValueError("X and Y spacing is not the same. Cannot write *.asc file in the specified format.")
        head = [("NCOLS", ncols), ("NROWS", nrows), ("XLLCORNER", x[0] - dx / 2.0), ("YLLCORNER", y[0] - dy / 2.0),
                ("CELLSIZE", dx), ("NODATA_VALUE", no_data)]
    else:
        raise ValueError("style kwarg must be either 1 or 2.")
    lines = ["%-15s%-10s\n" % (key, ("%d" % val) if key in ("NCOLS", "NROWS") else ("%.2f" % val)) for key, val in head]
    cells = np.char.ljust(np.char.mod("%.2f", z[::-1].astype(float)), 16)  # northernmost row first, 16-wide left-justified
    body = "\n".join("".join(row) for row in cells)
    with io.open(filename, "w") as f:
        f.write("".join(lines) + body)


_ASC_KEYS = {"ncols": "ncols", "nrows": "nrows", "xllcorner": "xllcorner", "xllcenter": "xllcenter", "yllcorner": "yllcorner",
             "yllcenter": "yllcenter", "cellsize": "cellsize", "cell_size": "cellsize", "dx": "dx", "dy": "dy",
             "nodata_value": "no_data", "nodatavalue": "no_data"}


def read_asc_grid(filename, footer=0):
    """Read an ARC ASCII grid -> (grid (M, N) oriented like X-Y space, x (N,), y (M,), cellsize or (dx, dy), no_data)."""
    h = {}
    nhead = 0

    def complete():
        origin = ("xllcorner" in h and "yllcorner" in h) or ("xllcenter" in h and "yllcenter" in h)
        step = "cellsize" in h or ("dx" in h and "dy" in h)
        return "ncols" in h and "nrows" in h and origin and step and "no_data" in h

    with io.open(filename, "r") as f:
        while not complete():
            key, value = f.readline().split()
            nhead += 1
            if key.lower() not in _ASC_KEYS:
                raise IOError("could not read *.asc file. Error in header.")
            name = _ASC_KEYS[key.lower()]
            h[name] = int(value) if name in ("ncols", "nrows") else float(value)
    grid = np.flipud(np.genfromtxt(filename, skip_header=nhead, skip_footer=footer))
    ncols, nrows = h["ncols"], h["nrows"]
    if grid.shape[0] != nrows or grid.shape[1] != ncols:
        raise IOError("Error reading *.asc file. Encountered problem with header: NCOLS and/or NROWS does not match "
                      "number of columns/rows in data file body.")
    both = "dx" in h and "dy" in h
    sx, sy = (h["dx"], h["dy"]) if both else (h["cellsize"], h["cellsize"])
    if "xllcorner" in h and "yllcorner" in h:  # corner given: move to the cell centre
        x0, y0 = h["xllcorner"] + sx / 2.0, h["yllcorner"] + sy / 2.0
    else:
        x0, y0 = h["xllcenter"], h["yllcenter"]
    x = np.arange(x0, x0 + ncols * sx, sx)[:ncols]  # arange may overshoot by one through rounding
    y = np.arange(y0, y0 + nrows * sy, sy)[:nrows]
    return grid, x, y, h.get("cellsize", (h.get("dx"), h.get("dy"))), h["no_data"]


def _zmap_field(v, no_data):
    """One node of a ZMAP+ body: right-justified, 15 wide (14 for |v| >= 1e100), NaN -> no_data."""
    if np.isnan(v):
        text, tail = format(no_data, "13.7E"), 2
    elif abs(v) >= 1e100:
        text, tail = format(v, "13.7E"), 1
    elif abs(v) >= 1e6:
        text, tail = format(v, "13.7E"), 2
    else:
        text, tail = "%.4f" % v, 2
    text = text.strip()
    return text.rjust(max(13, len(text)) + tail)


def write_zmap_grid(x, y, z, filename="output.zmap", no_data=-999.0, coord_sys="<null>"):
    """Write z (M, N) at the cell centres x, y as a ZMAP+ grid: column by column, north to south, 5 nodes per line."""
    per_line, width = 5, 15
    x, y, z = _plain_grid(x, y, z, no_data)
    nx, ny = len(x), len(y)
    dx, dy = _spacing(x, y)
    x0, y0 = x[0], y[0]
    now = datetime.datetime.now()
    head = ["!", "!     ZIMS FILE NAME :  " + os.path.basename(filename),
            "!     FORMATTED FILE CREATION DATE: " + now.strftime("%d/%m/%Y"),
            "!     FORMATTED FILE CREATION TIME: " + now.strftime("%H:%M:%S"),
            "!     COORDINATE REFERENCE SYSTEM: " + coord_sys, "!",
            "@Grid HEADER, GRID, %d" % per_line,
            " %d, %s,  , 1 , 1" % (width, no_data),
            "   %d,  %d,  %s,  %s,  %s,  %s" % (ny, nx, x0, x0 + (nx - 1) * dx, y0, y0 + (ny - 1) * dy),
            "   %s,  0.0,  0.0    " % dx, "@"]
    out = [line + "\n" for line in head]
    for n in range(z.shape[1]):
        column = [_zmap_field(v, no_data) for v in z[::-1, n]]
        for lo in range(0, len(column), per_line):
            out.append("".join(column[lo:lo + per_line]) + "\n")
    with io.open(filename, "w") as f:
        f.write("".join(out))


def read_zmap_grid(filename):
    """Read a ZMAP+ grid -> (z (M, N), x (N,), y (M,), (dx, dy), no_data, coordinate system name)."""
    coord_sys, header, values = "<null>", [], []
    section = 0  # 0 before the first '@', 1 inside the header block, 2 in the body
    with io.open(filename, "r") as f:
        for line in f:
            if line.startswith("!"):
                if "COORDINATE REFERENCE SYSTEM" in line.split(":")[0]:
                    coord_sys = line.split(":")[1].replace("\n", "")
                continue
            tokens = [t.replace(",", "") for t in line.split()]
            if not tokens:
                break
            if tokens[0].startswith("@"):
                section += 1
                if section == 1:
                    header.append(tokens)
                continue
            if section == 1:
                header.append(tokens)
            elif section == 2:
                values.extend(float(t) for t in tokens)
    no_data = float(header[1][1])
    ny, nx = int(header[2][0]), int(header[2][1])
    x0, x1, y0, y1 = (float(t) for t in header[2][2:6])
    if nx * ny != len(values):
        raise IOError("Error reading *.zmap file. Encountered problem with header: (nx * ny) does not match with the "
                      "number items in data file body.")
    z = np.asarray(values).reshape(nx, ny).T[::-1]  # columns north-to-south in the file
    dx, dy = (x1 - x0) / (nx - 1), (y1 - y0) / (ny - 1)
    return z, np.arange(x0, x0 + nx * dx, dx), np.arange(y0, y0 + ny * dy, dy), (dx, dy), no_data, coord_sys


def space_back_to_front(string):
    """kriging_tools.py:462-464 (a helper for ZMAP header fields): a blank-padded token comes back right-aligned -- the text around
    the token (the token = the string without its blanks, taken as ONE separator) followed by the token.  Same corner cases as
    the reference: a string whose non-blank characters are not contiguous is returned with the token appended, an all-blank
    string raises ValueError (empty separator)."""
    token = string.replace(" ", "")
    return "".join(string.rsplit(token)) + token
