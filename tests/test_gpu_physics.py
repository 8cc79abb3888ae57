"""The known-answer physics checks of tests/physics_cases.py on the HIP path (through the C ABI, one environment): the
kernels are held to physics itself — free fall, momentum, resting contact, Coulomb friction, pendulum period, motor laws —
with the same tolerances as the CPU oracle (tests/test_oracle_physics.py), not only to the oracle."""
import pytest

import physics_cases as pc
from conftest import write_skeleton

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def test_known_answer_free_fall(tmp_path):
    pc.check_free_fall(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path)))


def test_known_answer_momentum_of_a_free_spinning_box(tmp_path):
    drift = pc.check_momentum_free_spinning_box(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path, scale=(0.1, 0.2, 0.3))))
    print("angular momentum drift over 600 steps: %.3g" % drift)


def test_known_answer_resting_box_has_four_contact_points(tmp_path):
    pc.check_resting_box(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path)))


def test_known_answer_sliding_box_decelerates_at_mu_g(tmp_path):
    first, dist, ideal = pc.check_sliding_friction(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path, scale=(0.5, 0.1, 0.5))))
    print("first sliding step loses %.6f m/s (mu g dt = %.6f); stops after %.3f m (ideal %.3f m)" % (first, 0.25 * pc.G * pc.DT, dist, ideal))


def test_known_answer_hinge_pendulum_period(tmp_path):
    skel, base_y = pc.skel_pendulum(write_skeleton, tmp_path)
    period, pred = pc.check_pendulum_period(pc.HipWorld(skel), base_y)
    print("pendulum period %.4f s, predicted %.4f s" % (period, pred))


def test_known_answer_slider_motor_reaches_target_velocity(tmp_path):
    skel = pc.skel_two_masses(write_skeleton, tmp_path, mass=0.25, force=1.0e6, name="motor_free.skel")
    pc.check_motor_reaches_target_velocity(pc.HipWorld(skel))


def test_known_answer_slider_motor_saturates_at_64_newton(tmp_path):
    skel = pc.skel_two_masses(write_skeleton, tmp_path, mass=1000.0, force=64.0, name="motor_sat.skel")
    rel, pred = pc.check_motor_saturates_at_max_force(pc.HipWorld(skel))
    print("relative velocity %.5f m/s, predicted %.5f m/s" % (rel, pred))


def test_known_answer_welded_pair_moves_as_one_body(tmp_path):
    drift, angle = pc.check_welded_pair_moves_as_one_body(pc.HipWorld(pc.skel_welded_pair(write_skeleton, tmp_path)))
    print("fixed constraint after 240 steps: relative position drift %.2e m, relative rotation %.2e rad" % (drift, angle))


def test_known_answer_hinge_limit_holds(tmp_path):
    skel, base_y = pc.skel_limited_pendulum(write_skeleton, tmp_path)
    worst = pc.check_hinge_limit_holds(pc.HipWorld(skel), base_y)
    print("largest swing angle %.4f rad against a 0.3 rad limit" % worst)


def test_known_answer_impact_does_not_bounce(tmp_path):
    res = pc.check_impact_does_not_bounce(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path)))
    for v, up, low in res:
        print("impact at %.2f m/s: largest upward velocity afterwards %.4f m/s, deepest point %.4f m below rest" % (v, up, -low))


def test_known_answer_static_friction_holds(tmp_path):
    left, kick = pc.check_static_friction_holds(pc.HipWorld(pc.skel_cube(write_skeleton, tmp_path, scale=(0.5, 0.1, 0.5))))
    print("sideways velocity one step after a %.5f m/s kick: %.2e m/s" % (kick, left))


def test_known_answer_slider_stops_at_its_limits(tmp_path):
    skel = pc.skel_two_masses(write_skeleton, tmp_path, mass=0.25, force=1.0e6, name="motor_limits.skel")
    hi, lo = pc.check_slider_limits(pc.HipWorld(skel))
    print("slider length between %.4f and %.4f m (limits 0 and 2 m)" % (lo, hi))


def test_known_answer_hinge_removes_off_axis_rotation(tmp_path):
    skel, base_y = pc.skel_pendulum(write_skeleton, tmp_path)
    left, tilt = pc.check_hinge_removes_off_axis_rotation(pc.HipWorld(skel), base_y)
    print("off-axis relative spin after one step %.2e rad/s; hinge axes 1 - cos(angle) <= %.1e over 120 steps" % (left, tilt))
