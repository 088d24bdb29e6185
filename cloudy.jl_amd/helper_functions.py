"""Index maps and normalising factors of the flat moment vector (src/helper_functions.jl).

These define the plane order of every batched array handed to libcloudy_hip.so.
"""


def get_dist_moment_ind(NProgMoms, i, m):
    """helper_functions.jl:13-20 (1-based i, m as in the reference)."""
    if not (0 < i <= len(NProgMoms)):
        raise IndexError("distribution index out of range")
    if not (0 < m <= NProgMoms[i - 1]):
        raise ValueError("moment index must be positive integer and equal or smaller than the dist number of "
                         "prognostic moments!!!")
    return m if i == 1 else sum(NProgMoms[: i - 1]) + m


def get_dist_moments_ind_range(NProgMoms, i):
    """helper_functions.jl:29-32 -> range of 1-based indices."""
    if not (0 < i <= len(NProgMoms)):
        raise IndexError("distribution index out of range")
    last = 0 if i == 1 else sum(NProgMoms[: i - 1])
    return range(last + 1, last + NProgMoms[i - 1] + 1)


def get_moments_normalizing_factors(NProgMoms, norms):
    """helper_functions.jl:40-53: norms[1] * norms[2]^(j-1), mode-major."""
    if norms[0] <= 0 or norms[1] <= 0:
        raise ValueError("norms must be positive!")
    return tuple(norms[0] * norms[1] ** (j - 1) for n in NProgMoms for j in range(1, n + 1))


def rflatten(tup):
    """helper_functions.jl:55-58."""
    out = []
    for t in tup:
        if isinstance(t, (tuple, list)):
            out.extend(rflatten(t))
        else:
            out.append(t)
    return tuple(out)
