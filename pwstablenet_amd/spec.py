"""Static description of the generator's parameters and layer graph (pure Python, no torch).

Mirrors the module registration order of the reference generator
(reference lib/networks_cascading.py:109-149) so that ``state_dict()`` keys, shapes and order are
identical: conv weights are OIHW ``(Cout, Cin, k, k)``, transposed-conv weights are IOHW
``(Cin, Cout, k, k)``.  ``down_bottom`` registers ``conv_same`` before ``mpconv`` (reference :274-287),
``up_bottom`` registers ``mpconv`` before ``conv_same`` (reference :334-341).
"""
from collections import namedtuple

# kind: 'conv' (nn.Conv2d) or 'convT' (nn.ConvTranspose2d); k/s/p = kernel/stride/padding
LayerSpec = namedtuple("LayerSpec", "name kind cin cout k s p")


def layer_specs(input_nc=31, output_nc=2, ngf=64):
    """All 46 conv / transposed-conv layers in state-dict order."""
    g = ngf
    L = []

    def conv(name, cin, cout, k=3, s=2, p=1):
        L.append(LayerSpec(name, "conv", cin, cout, k, s, p))

    def convT(name, cin, cout, k, s, p):
        L.append(LayerSpec(name, "convT", cin, cout, k, s, p))

    # stage 1 encoder (reference :112-119)
    conv("transfer.mpconv.0", input_nc, g, 5, 1, 2)
    enc = [(g, g), (g, 2 * g), (2 * g, 4 * g), (4 * g, 4 * g), (4 * g, 4 * g), (4 * g, 4 * g), (4 * g, 4 * g)]
    for i, (ci, co) in enumerate(enc, 1):
        conv("down%d.mpconv.0" % i, ci, co)
    # stage 1 decoder up7..up1 (reference :121-127)
    dec = {7: (4 * g, 4 * g), 6: (8 * g, 4 * g), 5: (8 * g, 4 * g), 4: (8 * g, 4 * g), 3: (8 * g, 2 * g),
           2: (4 * g, g), 1: (2 * g, g)}
    for lvl in range(7, 0, -1):
        convT("up%d.mpconv.0" % lvl, dec[lvl][0], dec[lvl][1], 4, 2, 1)
    conv("out.mpconv.0", g, output_nc, 3, 1, 1)  # reference :128
    # stage 2/3 encoder down_bottom1..7 (reference :130-136)
    for i, (ci, co) in enumerate(enc, 1):
        conv("down_bottom%d.conv_same.0" % i, ci, ci, 3, 1, 1)
        conv("down_bottom%d.mpconv.0" % i, ci if i == 1 else 2 * ci, co)
    # stage 2/3 decoder up_bottom7..1: (input_nc, output_nc, inner_nc) (reference :140-146)
    ub = {7: (4 * g, 4 * g, 8 * g), 6: (8 * g, 4 * g, 16 * g), 5: (8 * g, 4 * g, 16 * g),
          4: (8 * g, 4 * g, 16 * g), 3: (8 * g, 2 * g, 16 * g), 2: (4 * g, g, 8 * g), 1: (2 * g, g, 4 * g)}
    for lvl in range(7, 0, -1):
        ci, co, inner = ub[lvl]
        convT("up_bottom%d.mpconv.0" % lvl, inner, co, 4, 2, 1)
        convT("up_bottom%d.conv_same.0" % lvl, ci, ci, 3, 1, 1)
    # affine head (reference :148-149)
    conv("flatten.mpconv.0", 4 * g, 8 * g, 2, 1, 0)
    conv("linear.mpconv.0", 8 * g, 6, 1, 1, 0)
    return L


def weight_shape(ls):
    if ls.kind == "conv":
        return (ls.cout, ls.cin, ls.k, ls.k)
    return (ls.cin, ls.cout, ls.k, ls.k)


def param_specs(input_nc=31, output_nc=2, ngf=64):
    """[(key, shape)] for all 92 tensors, in state-dict order (no ``module.`` prefix)."""
    out = []
    for ls in layer_specs(input_nc, output_nc, ngf):
        out.append((ls.name + ".weight", weight_shape(ls)))
        out.append((ls.name + ".bias", (ls.cout,)))
    return out


def num_params(input_nc=31, output_nc=2, ngf=64):
    n = 0
    for _, shp in param_specs(input_nc, output_nc, ngf):
        k = 1
        for d in shp:
            k *= d
        n += k
    return n
