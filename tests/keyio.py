"""Reader for the .key text format (test helper)."""
import numpy as np


def read_key(path_or_file):
    f = open(path_or_file, "rt") if isinstance(path_or_file, str) else path_or_file
    header, rows, count = [], [], None
    for line in f:
        if line.startswith("#"):
            header.append(line.rstrip("\n"))
        elif line.startswith("Features:"):
            count = int(line.split(":")[1])
        elif line.startswith("Scale-space"):
            columns = line.rstrip("\n")
        else:
            v = line.rstrip("\n").split("\t")
            if len(v) >= 81:
                rows.append([float(x) for x in v[:81]])
    f.close()
    return {"header": header, "count": count, "rows": np.array(rows, np.float64).reshape(-1, 81)}
