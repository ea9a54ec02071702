"""`asrtool` for the MI355X path: point cloud PLY in, triangle mesh PLY out -- the command line of the
reference (cpp/bin/main.cpp:114-177: `asrtool --in point_cloud.ply --out mesh.ply`, `--version`,
`--third-party-notices`) on top of adaptivesurfacereconstruction.reconstruct_surface.

    python adaptive-surface-reconstruction_amd/asrtool.py --in scan.ply --out mesh.ply [--weights model.pt]

The reference bundles its network as <resource dir>/model.pt (cpp/lib/asr.cpp:138-139); here the weights come
from --weights (a TorchScript archive with the same tensor names, a pickled state dict or an .npz) or from
$ASR_RESOURCE_DIR/{model_weights.npz, model_weights.pt, model.pt}.
"""
import os
import sys

HELP = """usage: asrtool --in point_cloud.ply --out mesh.ply

Arguments:
    in      Input point cloud with normal information in PLY format.
    out     Output mesh in PLY format.

Options:
    --weights FILE  Network weights (TorchScript model.pt, state dict .pt or .npz); default $ASR_RESOURCE_DIR
    --version  Prints the version information
    --third-party-notices  Prints third-party software notices
"""


def _option(argv, name):
    """value following `name` (Open3D's GetProgramOptionAsString, main.cpp:150-153) or None"""
    if name in argv:
        i = argv.index(name)
        if i + 1 < len(argv):
            return argv[i + 1]
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if "--version" in argv:
        import adaptivesurfacereconstruction as asr
        print("asrtool version " + asr.get_version_str())
        return 0
    if "--third-party-notices" in argv:
        import adaptivesurfacereconstruction as asr
        print(asr.get_third_party_notices())
        return 0
    inp, out = _option(argv, "--in"), _option(argv, "--out")
    if inp is None or out is None:
        sys.stdout.write(HELP)
        return 1
    import adaptivesurfacereconstruction as asr
    from asr_hip import ply
    print("reading points")
    points, normals, radii = ply.read_points(inp)
    print("%d / %d" % (len(points), len(points)))
    result = asr.reconstruct_surface(points, normals, radii, weights=_option(argv, "--weights"))
    ply.write_mesh(out, result["vertices"], result["triangles"])
    print("wrote %s: %d vertices, %d triangles" % (out, len(result["vertices"]), len(result["triangles"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
