"""Run the reference's OWN entry scripts on the MI355X render path, without editing a line of them:

    python -m intrinsicnerf_amd.launch  <IntrinsicNeRF>/object_level/run_nerf.py  --config configs/chair.txt [...]
    python -m intrinsicnerf_amd.launch  <IntrinsicNeRF>/train_SSR_main.py  --config_file SSR/configs/SSR_room0_config.yaml [...]

The reference is pure Python: the "operator interface" of its render path is a handful of module-level functions and three
trainer methods.  The launcher loads the script as a module (everything above its
``if __name__ == '__main__':`` block), rebinds exactly those names to this package's mirrors - in the namespaces the
reference's own code looks them up in - and then executes the script's own main block (seeds, default tensor type,
``train()``) as if it had been started with ``python``.  What INTEGRATION.md section A describes as two edits, done at run time.

  object_level/run_nerf.py    (run_nerf.py:32-139, 359-528; run_nerf_helpers.py:195-445)
      run_network, batchify_rays, render, render_rays, raw2outputs, NeRF, get_embedder, Embedder, sample_pdf, get_rays,
      get_rays_np, ndc_rays  [+ render_path, to8b with --inerf-render-path]
      ``create_nerf`` stays the reference's: its ``network_query_fn`` lambda is recognised structurally
      (object_level._as_network_query) once ``run_network`` in the script's namespace is this package's.
  train_SSR_main.py           (SSR/training/trainer.py:693-846; SSR/models/*.py; SSR/training/cluster.py:73-98)
      SSRTrainer.render_rays / volumetric_rendering / create_ssr <- ssr.SSRRenderMixin's; run_network, raw2outputs,
      sample_pdf, create_rays, Semantic_NeRF, get_embedder in every SSR module that holds them; the two cluster lookups.

``prepare(script)`` does everything but run the main block and returns the module (used by the tests).
"""
import ast
import importlib
import os
import sys
import types

OBJECT_SYMBOLS = ("run_network", "batchify_rays", "render", "render_rays", "raw2outputs", "NeRF", "get_embedder", "Embedder",
                  "sample_pdf", "get_rays", "get_rays_np", "ndc_rays")
OBJECT_OPTIONAL = ("render_path", "to8b")
SSR_METHODS = ("render_rays", "volumetric_rendering", "create_ssr")
SSR_SYMBOLS = ("run_network", "raw2outputs", "sample_pdf", "create_rays", "Semantic_NeRF", "get_embedder", "Embedder")
SSR_MODULES = ("SSR.training.trainer", "SSR.models.model_utils", "SSR.models.rays", "SSR.models.semantic_nerf")


def _split_main(source, filename):
    """(code of everything but the ``if __name__ == '__main__':`` blocks, code of their bodies) of a script."""
    tree = ast.parse(source, filename)
    body, main = [], []
    for node in tree.body:
        is_main = (isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name)
                   and node.test.left.id == "__name__" and len(node.test.comparators) == 1
                   and isinstance(node.test.comparators[0], ast.Constant) and node.test.comparators[0].value == "__main__")
        (main if is_main else body).append(node)
    main_body = [stmt for node in main for stmt in node.body]
    mk = lambda nodes: compile(ast.fix_missing_locations(ast.Module(body=nodes, type_ignores=[])), filename, "exec")
    return mk(body), mk(main_body)


def _kind(script):
    name = os.path.basename(script)
    if name == "run_nerf.py":
        return "object"
    if name == "train_SSR_main.py":
        return "ssr"
    raise SystemExit(f"intrinsicnerf_amd.launch: {name}: expected the reference's object_level/run_nerf.py or train_SSR_main.py")


def rebind_object_level(namespace, with_render_path=False):
    """The object-level mirrors into ``namespace`` (a module's ``__dict__``): returns the names it bound."""
    from . import object_level
    names = OBJECT_SYMBOLS + (OBJECT_OPTIONAL if with_render_path else ())
    for name in names:
        namespace[name] = getattr(object_level, name)
    if with_render_path and "Cluster_Manager" in namespace:
        # run_nerf.py:818,1071 call render_path(update_cluster=True): the mean-shift fitting stays the reference's own class
        # (run_nerf.py:24, :218), handed to the mirror as its factory
        import functools
        namespace["render_path"] = functools.partial(object_level.render_path, cluster_manager_factory=namespace["Cluster_Manager"])
    return names


def rebind_ssr(with_render_path=False):
    """The SSR mirrors into the (already imported) reference modules; returns {module name: [names bound]}."""
    from . import cluster as inerf_cluster, ssr
    bound = {}
    trainer = sys.modules["SSR.training.trainer"]
    methods = SSR_METHODS + (("render_path",) if with_render_path else ())
    for name in methods:
        setattr(trainer.SSRTrainer, name, getattr(ssr.SSRRenderMixin, name))
    for extra in ("return_raw", "check_numerics", "_staged"):          # what the mixin's methods read besides the trainer's attributes
        setattr(trainer.SSRTrainer, extra, getattr(ssr.SSRRenderMixin, extra))
    if with_render_path and hasattr(trainer, "Cluster_Manager"):
        # trainer.py:1065 renders with update_cluster = not self.no_cluster: the fitting is the reference's class (trainer.py:16, :1416-1418)
        trainer.SSRTrainer.cluster_manager_factory = staticmethod(trainer.Cluster_Manager)
    bound["SSR.training.trainer.SSRTrainer"] = list(methods)
    for mod_name in SSR_MODULES:
        mod = sys.modules.get(mod_name)
        if mod is None:
            continue
        here = [n for n in SSR_SYMBOLS if hasattr(mod, n)]
        for n in here:
            setattr(mod, n, getattr(ssr, n))
        bound[mod_name] = here
    cl = sys.modules.get("SSR.training.cluster")
    if cl is not None and hasattr(cl, "Cluster_Manager"):               # cluster.py:73-98: one HIP launch for all classes
        cl.Cluster_Manager.dest_color = inerf_cluster.dest_color
        cl.Cluster_Manager.dest_class = inerf_cluster.dest_class
        bound["SSR.training.cluster.Cluster_Manager"] = ["dest_color", "dest_class"]
    return bound


def prepare(script, with_render_path=False):
    """Load the reference script as a module (without its main block), rebind the render path, return (module, main code)."""
    script = os.path.abspath(script)
    kind = _kind(script)
    root = os.path.dirname(script)
    if root not in sys.path:
        sys.path.insert(0, root)                       # what `python script.py` does: the script's directory leads sys.path
    with open(script, "r") as fh:
        body, main = _split_main(fh.read(), script)
    mod = types.ModuleType("run_nerf" if kind == "object" else "train_SSR_main")
    mod.__file__ = script
    sys.modules[mod.__name__] = mod
    exec(body, mod.__dict__)
    if kind == "object":
        mod.__dict__["__inerf_bound__"] = rebind_object_level(mod.__dict__, with_render_path)
    else:
        importlib.import_module("SSR.training.trainer")
        mod.__dict__["__inerf_bound__"] = rebind_ssr(with_render_path)
    return mod, main


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    with_render_path = "--inerf-render-path" in argv
    if with_render_path:
        argv.remove("--inerf-render-path")
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0
    script = argv[0]
    from . import _capi
    _capi.lib()                                        # fail now, and loudly, if the HIP library is missing
    mod, main_code = prepare(script, with_render_path)
    sys.argv = [script] + argv[1:]                     # the script's own argument parser sees its own command line
    mod.__dict__["__name__"] = "__main__"
    exec(main_code, mod.__dict__)
    return 0


if __name__ == "__main__":
    sys.exit(main())
