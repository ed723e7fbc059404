"""Writes tests/golden/*_state_dict_layout.txt: the ordered (key, shape) list of the backbone's checkpoint layout, from the
oracle (oracle/resnet_dilated_oracle.py).  tests/test_oracle.py holds the oracle, the product module and the independently
derived layout of tests/backbone_second_statement.py against these files.    python tests/golden/make_state_dict_layouts.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')); from oracle import resnet_dilated_oracle as orc
for arch, D in (("Resnet34_8s", 3), ("Resnet50_8s", 32)):
    m = orc.build(arch, D)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '%s_d%d_state_dict_layout.txt') % (arch.lower(), D)
    with open(path, 'w') as f:
        f.write("# ordered state_dict() layout of %s(num_classes=%d): <key> <shape>; %d tensors, %d parameters\n" % (
            arch, D, len(m.state_dict()), sum(p.numel() for p in m.parameters())))
        for k, v in m.state_dict().items():
            f.write("%s %s\n" % (k, "x".join(str(s) for s in v.shape) or "scalar"))
    print(path, len(m.state_dict()))
