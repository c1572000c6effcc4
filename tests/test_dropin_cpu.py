"""CPU: the Python seam of the drop-in boundary: the reference's dotted class names resolve to this package after
``holoscene_amd.dropin.install()``, and the model builds from the reference's stock configuration as
parsed by holoscene_amd.utils.conf (fixture tests/golden/stock_conf_parsed.json, generated from the reference's own .conf by
tests/golden/make_golden.py::run_conf -- the file itself does not travel)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_CONF = "/root/reference/confs/replica/room_0/replica_room_0.conf"


def _run(code, cwd, extra_path=()):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([*extra_path, ROOT]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


# utils.general.get_class of the reference (utils/general.py:188-194), restated
GET_CLASS = """
def get_class(kls):
    parts = kls.split('.')
    m = __import__('.'.join(parts[:-1]))
    for comp in parts[1:]:
        m = getattr(m, comp)
    return m
"""


def test_dotted_names_resolve_and_stock_conf_builds_the_model(tmp_path):
    """No reference on the path (the GPU box, any user's site): the aliases are bound wholesale; model_class / loss_class of the
    stock conf resolve through the reference's get_class; HoloSceneNetwork(conf.model) has the 24 713 271 parameters of SURVEY 8."""
    out = _run(GET_CLASS + f"""
import json, sys
import holoscene_amd.dropin as dropin
from holoscene_amd.utils.conf import Conf
print(sorted(dropin.install().items()))
conf = Conf(json.load(open({os.path.join(GOLDEN, 'stock_conf_parsed.json')!r}))['replica_room_0'])
Net = get_class(conf.get_string('train.model_class'))
Loss = get_class(conf.get_string('train.loss_class'))
import holoscene_amd.model.network as n, holoscene_amd.model.loss as l
assert Net is n.HoloSceneNetwork and Loss is l.HoloSceneLoss
from hashencoder.hashgrid import HashEncoder            # the reference's own import line (model/network.py:10)
from model.ray_sampler import ErrorBoundSampler
from model.density import LaplaceDensity
import holoscene_amd.hashencoder.hashgrid as hg
assert HashEncoder is hg.HashEncoder
model = Net(conf=conf.get_config('model'), graph_node_dict=None, num_images=8)      # holoscene_train.py:137-143
loss = Loss(**conf.get_config('loss'))                                              # holoscene_train.py:148-151
from model.network import ObjectSDFNetwork, SingleObjectImplicitNetworkGrid      # holoscene_train_post.py:58
import holoscene_amd.model.object_network as on
assert ObjectSDFNetwork is on.ObjectSDFNetwork and get_class('model.network.SingleObjectImplicitNetworkGrid') is on.SingleObjectImplicitNetworkGrid
print('PARAMS', sum(p.numel() for p in model.parameters()))
print('DOUT', model.implicit_network.d_out, type(model.ray_sampler).__name__, model.ray_sampler.N_samples_eval)
""", cwd=str(tmp_path))
    assert "PARAMS 24713271" in out, out
    assert "DOUT 32 ErrorBoundSampler 128" in out, out
    assert "('model.network', 'replaced')" in out


def test_overlay_keeps_the_reference_modules_other_symbols(tmp_path):
    """Inside a reference checkout (simulated by a stand-in ``model`` package on the path): mirrored classes are overlaid on the
    reference's modules, everything else the trainers import from them stays the reference's."""
    pkg = tmp_path / "model"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "loss.py").write_text("def compute_scale_and_shift(*a):\n    return 'reference helper'\n\nclass HoloSceneLoss:\n    pass\n")
    (pkg / "network.py").write_text("from hashencoder.hashgrid import HashEncoder\n\nclass HoloSceneNetwork:\n    pass\n\nclass ColorImplicitNetworkSingle:\n    pass\n")
    for name in ("ray_sampler", "density", "embedder"):
        (pkg / f"{name}.py").write_text("")
    out = _run(GET_CLASS + """
import holoscene_amd.dropin as dropin
done = dropin.install()
import model.loss, model.network
import holoscene_amd.model.loss as l, holoscene_amd.model.network as n, holoscene_amd.hashencoder.hashgrid as hg
assert done['model.loss'] == 'overlaid' and done['hashencoder.hashgrid'] == 'replaced', done
assert model.loss.compute_scale_and_shift() == 'reference helper'
assert get_class('model.loss.HoloSceneLoss') is l.HoloSceneLoss
assert get_class('model.network.HoloSceneNetwork') is n.HoloSceneNetwork
assert model.network.ColorImplicitNetworkSingle.__module__ == 'model.network'       # not mirrored: stays the reference's
assert model.network.HashEncoder is hg.HashEncoder                                  # its own import line got this build's encoder
print('OK')
""", cwd=str(tmp_path), extra_path=[str(tmp_path)])
    assert "OK" in out


def test_reader_handles_the_syntax_of_the_stock_conf():
    from holoscene_amd.utils.conf import parse_string
    c = parse_string('''
        train{ expname = a_b  # comment
            learning_rate = 5.0e-4
            model_class = model.network.HoloSceneNetwork }
        plot{ grid_boundary = [-1.0, 1.0] }   // another comment
        model{ use_bg_reg = True
            implicit_network { dims = [256, 256]
                skip_in = [4] }
            density { params_init{ beta = 0.1 } } }
        quoted = "a # not a comment"
    ''')
    assert c.get_float("train.learning_rate") == 5e-4 and c.get_string("train.model_class") == "model.network.HoloSceneNetwork"
    assert c.get_list("plot.grid_boundary") == [-1.0, 1.0] and c.get_bool("model.use_bg_reg") is True
    assert c.get_config("model.implicit_network").get_list("dims") == [256, 256]
    assert c.get_float("model.density.params_init.beta") == 0.1 and c.get_string("quoted") == "a # not a comment"
    with pytest.raises(KeyError):
        c.get_int("train.missing")
    assert c.get_int("train.missing", default=7) == 7


@pytest.mark.skipif(not os.path.exists(REF_CONF), reason="the reference checkout exists in the build container only")
def test_committed_conf_fixture_is_what_the_reference_file_parses_to():
    from holoscene_amd.utils.conf import parse_file
    want = json.load(open(os.path.join(GOLDEN, "stock_conf_parsed.json")))["replica_room_0"]
    assert json.loads(json.dumps(parse_file(REF_CONF))) == want
