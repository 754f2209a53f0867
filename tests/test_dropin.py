"""The drop-in boundary as the reference's scripts see it: with ``dropin/`` first on ``sys.path`` the literal import
statements of train.py:13, train_coarse.py:7 and eval.py:24-26 must resolve — the mirrored names to the MI355X
implementation, the un-mirrored submodules (models.mano, models.inception, ...) to a reference checkout later on the
path.  The reference does not travel, so its side is a stand-in directory created by the test; names are hard-coded."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT, *extra_path])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env,
                       cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_reference_import_statements_resolve(tmp_path):
    # a stand-in for the reference checkout: only the submodules the mirror does NOT provide
    ref = tmp_path / "SCAT"
    (ref / "models" / "helper").mkdir(parents=True)
    (ref / "models" / "__init__.py").write_text("")
    (ref / "models" / "helper" / "__init__.py").write_text("")
    (ref / "models" / "mano.py").write_text("class ManoHand: pass\ndef rot_pose_beta_to_mesh(*a): return 'ref'\n")
    (ref / "models" / "inception.py").write_text("class Inception3: origin = 'reference'\n")
    (ref / "models" / "hand_net.py").write_text("raise ImportError('the reference hand_net must be shadowed')\n")
    out = _run("""
        from models.hand_net import EncoderTransformer                                    # train.py:13
        from models.hand_net import EncoderTransformer, EncoderTransformerCoarse, EncoderTransformerHRNet, EncoderTransformerInception   # train_coarse.py:7
        from models.hand_net import EncoderTransformer, EncoderTransformerCoarse          # eval.py:24
        from models.mano import ManoHand,rot_pose_beta_to_mesh                            # eval.py:25
        from models.hand_net import H3DWEncoder                                           # eval.py:26
        import models.inception, models.helper, models.resnet, models.vit, models.hrnet
        import models.vision_transformer, models.vision_transformer_attn, models.vision_performer
        import scat_amd.models.hand_net as M
        assert EncoderTransformer is M.EncoderTransformer and H3DWEncoder is M.H3DWEncoder
        assert EncoderTransformerCoarse is M.EncoderTransformerCoarse and EncoderTransformerHRNet is M.EncoderTransformerHRNet
        assert rot_pose_beta_to_mesh() == 'ref' and models.inception.Inception3.origin == 'reference'
        assert models.resnet.resnet50 is __import__('scat_amd.models.resnet', fromlist=['x']).resnet50
        for name in ('get_model', 'PositionalEncoding'):
            assert hasattr(models.hand_net, name), name
        try:
            EncoderTransformerInception(None, None)
        except NotImplementedError as e:
            assert 'hand_net.py:87-146' in str(e)
        else:
            raise AssertionError('EncoderTransformerInception must refuse construction')
        print('ok')
    """, extra_path=[str(ref)])
    assert out.strip().endswith("ok")


def test_mirror_alone_on_the_path():
    """without a reference checkout behind it the mirrored modules still import (GPU box situation)"""
    out = _run("""
        from models.hand_net import EncoderTransformer, EncoderTransformerCoarse, EncoderTransformerHRNet, EncoderTransformerInception, H3DWEncoder
        from models.vit import Transformer
        from models.vision_performer import performer_attn_block, ViP
        from models.hrnet import HRNet
        try:
            import models.mano
        except ModuleNotFoundError:
            print('ok')
    """)
    assert out.strip().endswith("ok")
