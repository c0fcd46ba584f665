"""Problem builders for the BASELINE configurations (C1..C5).

Every builder takes an ``api`` namespace (see :func:`load_api`) so that the very
same construction code can run on this repository's ``mpc_interface`` mirror
and -- in the build container only, from ``tests/golden/make_golden.py`` -- on
the real reference package, which is how the golden vectors are produced.

C1/C2 ``biped``     : counterpart of the reference example
                      use_examples/simple_functional_example/biped_formulation.py:23-191
C3    ``lipm3d``    : 3-D LIPM, CoP box + height band + terminal box, N=32
C4    ``random_lti``: random stable LTI nx=12 nu=6, N=64
C5    ``ltv_lipm``  : per-step (A_k, B_k) for the fill kernel, N=100
"""
import importlib
import types

import numpy as np


def load_api(package="mpc_interface"):
    """Collect the description classes of ``package`` in one namespace."""
    mods = {
        name: importlib.import_module(package + "." + name)
        for name in ("body", "dynamics", "goal", "restrictions", "combinations", "tools")
    }
    return types.SimpleNamespace(
        Formulation=mods["body"].Formulation,
        ControlSystem=mods["dynamics"].ControlSystem,
        ExtendedSystem=mods["dynamics"].ExtendedSystem,
        DomainVariable=mods["dynamics"].DomainVariable,
        Cost=mods["goal"].Cost,
        Constraint=mods["restrictions"].Constraint,
        Box=mods["restrictions"].Box,
        LineCombo=mods["combinations"].LineCombo,
        tools=mods["tools"],
    )


# --------------------------------------------------------------------------
# C1 / C2: biped walking on a LIPM with jerk input
# --------------------------------------------------------------------------
class BipedConfig:
    """Constants of the walking example (reference biped_configuration.py:11-52).

    ``step_samples`` is the number of MPC periods per step; the horizon spans
    ``num_steps`` steps.  The shipped example has 12 (N=24); BASELINE's C1/C2
    use 8 (N=16).
    """

    def __init__(self, step_samples=8, num_steps=2, mpc_period=0.1):
        self.system = "J->CCC"
        self.num_steps = num_steps
        self.mpc_period = mpc_period
        self.step_samples = int(step_samples)
        self.horizon_lenght = self.num_steps * self.step_samples

        self.feet_sep = 0.08
        self.foot_corner = np.array([0.1, 0.05])
        self.stepping_corner = np.array([0.3, 0.1])
        self.stepping_center = np.array(
            [0, 2 * self.foot_corner[1] + self.stepping_corner[1] + self.feet_sep]
        )
        self.strt_y = self.feet_sep / 2 + self.foot_corner[1]
        self.com_height = 0.877
        self.gravity = [0, 0, -9.81]
        self.omega = np.sqrt(-self.gravity[2] / self.com_height)

        self.cop_safety_margin = 0.02
        self.target_vel = np.array([0.6, 0, 0])
        self.cost_weights = {
            "minimize jerk": 0.001,
            "track velocity": 0.01,
            "relax ankles": 1,
            "terminal": 0,
        }


def biped(api, conf=None, reduced=False):
    """Walking formulation: steps + LIPM + bias, 3 boxes, 6 costs.

    ``reduced``: without the zero-weight terminal cost and the terminal box of the example
    (biped_formulation.py:95-105, 129-131) -- the "3 costs, 2 box constraints" BASELINE.json's
    north star counts (cost kinds: relax ankles, minimize jerk, track velocity).

    Domain per axis: ``Ds`` (next step displacements, width changes with the
    walking phase), ``s0`` (current support), ``CoM_dddot`` (jerk, unknown),
    ``x0`` (CoM state), ``n`` (CoP bias).  Unknowns:
    ``[CoM_dddot_x, Ds_x, CoM_dddot_y, Ds_y]``.
    """
    conf = conf or BipedConfig()
    t = api.tools
    N, axes, w = conf.horizon_lenght, ["_x", "_y"], conf.omega

    # support-foot position along the horizon: s = F s0 + E Ds
    steps = api.ExtendedSystem(
        "Ds", "s", "s",
        S=np.ones([N, 1]),
        U=t.plan_steps(N, 0, regular_time=conf.step_samples),
        axes=axes,
        how_to_update_matrices=t.update_step_matrices,
        time_variant=True,
    )
    steps.define_output("stamps", {"s0": 1, "Ds": 1}, time_variant=True,
                        how_to_update=t.adapt_size)

    pendulum = api.ControlSystem.from_name(conf.system, axes, tau=conf.mpc_period, omega=w)
    lip = api.ExtendedSystem.from_cotrol_system(pendulum, "x", N)
    bias = api.DomainVariable("n", N, axes)

    shift_one = np.diag(np.ones([N - 1]), 1)      # bias acts one sample later
    outputs = {}
    for a in axes:
        outputs["b" + a] = api.LineCombo({"CoM" + a: 1, "CoM_ddot" + a: -1 / w**2})
        outputs["(b+n-s)" + a] = api.LineCombo(
            {"b" + a: 1, "n" + a: shift_one, "s" + a: -1})
        outputs["(c-s)" + a] = api.LineCombo({"CoM" + a: 1, "s" + a: -1})
        outputs["DCM" + a] = api.LineCombo({"CoM" + a: 1, "CoM_dot" + a: 1 / w})
        outputs["(DCM-s)" + a] = api.LineCombo({"DCM" + a: 1, "s" + a: -1})

    foot = t.make_simetric_vertices(conf.foot_corner)
    last = range(N - 1, N)

    support_polygon = api.Box.task_space("(b+n-s)", foot, axes)
    stepping_area = api.Box.task_space(
        "Ds", t.make_simetric_vertices(conf.stepping_corner), axes,
        how_to_update=t.update_stepping_area, time_variant=True)
    stepping_area.update(step_count=0, n_next_steps=steps.domain["Ds_x"],
                         xy_lenght=conf.stepping_center)
    terminal_box = api.Box.task_space("(DCM-s)", foot, axes, schedule=last)
    support_polygon.set_safety_margin(conf.cop_safety_margin)
    terminal_box.set_safety_margin(conf.cop_safety_margin)

    cw = conf.cost_weights
    goals = {
        "relax ankles x": api.Cost("(b+n-s)", cw["relax ankles"], aim=[0], axes=["_x"]),
        "relax ankles y": api.Cost("(b+n-s)", cw["relax ankles"], aim=[0], axes=["_y"]),
        "minimize jerk": api.Cost("CoM_dddot", cw["minimize jerk"], aim=[0, 0], axes=axes),
        "track vel_x": api.Cost("CoM_dot", cw["track velocity"],
                                aim=conf.target_vel[0], axes=["_x"]),
        "track vel_y": api.Cost("CoM_dot", cw["track velocity"],
                                aim=conf.target_vel[1], axes=["_y"]),
        "terminal_cost": api.Cost("(DCM-s)", cw["terminal"], aim=[0, 0], axes=axes,
                                  schedule=last),
    }

    form = api.Formulation()
    form.incorporate_dynamics("steps", steps)
    form.incorporate_dynamics("LIP", lip)
    form.incorporate_dynamics("bias", bias)
    form.incorporate_definitions(outputs)
    for name, goal in goals.items():
        if not (reduced and name == "terminal_cost"):
            form.incorporate_goal(name, goal)
    form.incorporate_box("stepping area", stepping_area)
    form.incorporate_box("support_polygon", support_polygon)
    if not reduced:
        form.incorporate_box("terminal_Constraint", terminal_box)

    form.identify_qp_domain(["CoM_dddot_x", "Ds_x", "CoM_dddot_y", "Ds_y"])
    form.make_preview_matrices()

    def per_tick(body, **kargs):
        """Needs ``step_times`` (array of next step instants) and
        ``step_count`` (steps done so far)."""
        body.dynamics["steps"].update(step_times=kargs["step_times"], N=N)
        body.constraint_boxes["stepping area"].update(
            step_count=kargs["step_count"],
            n_next_steps=body.dynamics["steps"].domain["Ds_x"],
            xy_lenght=conf.stepping_center,
        )

    form.set_updating_rule(per_tick)
    return form


class StepClock:
    """Step-time bookkeeping of the walking loop (reference
    biped_mpc_loop.py:37-45): ``step_times`` count down each tick and wrap by
    ``step_samples`` when the first reaches -1, incrementing ``step_count``."""

    def __init__(self, step_samples, n_steps):
        self.n = step_samples
        self.step_times = np.array([(i + 1) * self.n - 1 for i in range(n_steps)])
        self.step_count = 0

    def tick(self):
        self.step_times -= 1
        if self.step_times[0] == -1:
            self.step_times += self.n
            self.step_count += 1


def biped_given_collector(form, rng, bias_sigma=0.0):
    """One random set of given values for the biped."""
    collector = {}
    for var, ids in form.given_ID.items():
        size = len(ids)
        if var.startswith("x0"):
            collector[var] = rng.normal(0, 0.05, [size, 1])
        elif var.startswith("s0"):
            collector[var] = rng.uniform(-0.1, 0.1, [size, 1])
        elif bias_sigma:
            collector[var] = rng.normal(0, bias_sigma, [size, 1])
        else:
            collector[var] = np.zeros([size, 1])
    return collector


# --------------------------------------------------------------------------
# the 9-step case of the reference's own python/tests/test_body.py: exercises every
# branch of the cost / constraint assembly (L / no L, schedule, cross term, box)
# --------------------------------------------------------------------------
def body_case(api):
    """Same problem as reference python/tests/test_body.py:22-98, 124."""
    t = api.tools
    axes = ["_x", "_y"]
    lip = api.ControlSystem.from_name("J->CCC", tau=0.1, omega=3.5, axes=axes)
    lip_ext = api.ExtendedSystem.from_cotrol_system(lip, "x", 9)
    lip_ext.define_output("DCM", {"CoM": 1, "CoM_dot": 1 / 3.5})

    E = t.plan_steps(9, 1, step_times=np.array([2, 5, 8]))[:, :, None]
    steps = api.ExtendedSystem(["Ds"], ["s"], "s", np.ones([9, 1, 1]), [E], axes,
                               how_to_update_matrices=t.update_step_matrices,
                               time_variant=True)
    bias = api.DomainVariable("n", 9, axes)

    form = api.Formulation()
    form.incorporate_dynamics("steps", steps)
    form.incorporate_dynamics("LIP", lip_ext)
    form.incorporate_dynamics("n", bias)
    form.incorporate_definitions({
        "DCM_x": api.LineCombo({"CoM_x": 1, "CoM_dot_x": 1 / 3.5}),
        "DCM_y": api.LineCombo({"CoM_y": 1, "CoM_dot_y": 1 / 3.5}),
    })
    form.incorporate_constraint("kinematics", api.Constraint("CoM_x", 4))
    form.incorporate_constraint(
        "steppingArea", api.Constraint("s", 10, arrow=[1, 1], axes=axes, L=np.eye(9)))
    form.incorporate_goal("velocity", api.Cost("CoM_dot", aim=[1, 2], weight=10, axes=axes))
    form.incorporate_goal("stability", api.Cost("DCM_x", aim=2, weight=1))
    form.incorporate_goal(
        "terminal", api.Cost("CoM", aim=[50, 50], weight=100, axes=axes, schedule=range(8, 9)))
    form.incorporate_goal(
        "crossed", api.Cost("DCM_x", aim=2, weight=1, cross="s_y", cross_L=np.eye(9)))
    form.incorporate_box(
        "kine", api.Box.task_space("CoM", np.array([[0, 1], [1, 0], [0, -1], [-1, 0]]), axes))
    form.identify_qp_domain(["x0_x", "x0_y", "Ds_x", "Ds_y", "CoM_dddot_x", "CoM_dddot_y"])
    form.make_preview_matrices()
    return form


# --------------------------------------------------------------------------
# C3: 3-D LIPM, CoP box + height band + terminal DCM box
# --------------------------------------------------------------------------
def lipm3d(api, N=32, tau=0.1, omega=3.3445, foot_corner=(0.1, 0.05),
           z_band=(0.75, 0.95), target_vel=(0.3, 0.0)):
    """96 unknowns (jerk on x, y, z), 196 inequality rows at N=32."""
    t = api.tools
    axes = ["_x", "_y", "_z"]
    plane = ["_x", "_y"]

    pendulum = api.ControlSystem.from_name("J->CCC", axes, tau=tau, omega=omega)
    lip = api.ExtendedSystem.from_cotrol_system(pendulum, "x", N)
    lip.define_output("b", {"CoM": 1, "CoM_ddot": -1 / omega**2})
    lip.define_output("DCM", {"CoM": 1, "CoM_dot": 1 / omega})

    foot = t.make_simetric_vertices(np.array(foot_corner))
    cop_box = api.Box.task_space("b", foot, plane)
    terminal = api.Box.task_space("DCM", foot, plane, schedule=range(N - 1, N))
    height = [
        api.Constraint("CoM_z", z_band[1]),                             # z < z_max
        api.Constraint("CoM_z", 0.0, arrow=[-1], center=[z_band[0]]),   # -(z - z_min) < 0
    ]

    form = api.Formulation()
    form.incorporate_dynamics("LIP", lip)
    form.incorporate_goal("jerk", api.Cost("CoM_dddot", 1e-3, aim=[0, 0, 0], axes=axes))
    form.incorporate_goal("velocity",
                          api.Cost("CoM_dot", 1e-2, aim=list(target_vel), axes=plane))
    form.incorporate_goal("centre CoP", api.Cost("b", 1.0, aim=[0, 0], axes=plane))
    form.incorporate_constraint("height band", height)
    form.incorporate_box("CoP", cop_box)
    form.incorporate_box("terminal", terminal)
    form.identify_qp_domain(["CoM_dddot_x", "CoM_dddot_y", "CoM_dddot_z"])
    form.make_preview_matrices()
    return form


# --------------------------------------------------------------------------
# C4: random stable LTI system
# --------------------------------------------------------------------------
def random_lti_matrices(rng, nx=12, nu=6):
    """``A = 0.9 Q`` with Q orthogonal (QR of a normal matrix),
    ``B = normal / sqrt(nx)``."""
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    return 0.9 * Q, rng.standard_normal((nx, nu)) / np.sqrt(nx)


def random_lti(api, rng, nx=12, nu=6, N=64, bound=5.0):
    """State tracking + input effort, two-sided bounds on every state:
    ``no = nu N`` unknowns, ``2 nx N`` inequality rows."""
    A, B = random_lti_matrices(rng, nx, nu)
    inputs = ["u%d" % j for j in range(nu)]
    states = ["s%d" % i for i in range(nx)]
    system = api.ControlSystem(inputs, states, A, B)
    ext = api.ExtendedSystem.from_cotrol_system(system, "x", N)

    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    for name in states:
        form.incorporate_goal(
            "track " + name,
            api.Cost(name, float(rng.uniform(0.1, 1)), aim=[float(rng.normal())]))
    for name in inputs:
        form.incorporate_goal("effort " + name, api.Cost(name, float(rng.uniform(0.1, 1))))
    for name in states:
        form.incorporate_constraint(
            "bounds " + name,
            [api.Constraint(name, bound), api.Constraint(name, bound, arrow=[-1])])
    form.identify_qp_domain(inputs)
    form.make_preview_matrices()
    return form


# --------------------------------------------------------------------------
# C5: per-step dynamics -- for the fill kernel, and as an assembly
# --------------------------------------------------------------------------
def lipm_ltv(api, N=100, tau=0.1, omega=3.3445, foot_corner=(0.1, 0.05), target_vel=(0.3, 0.0)):
    """The C5-shaped assembly: the ``dP->CCC`` pendulum (input: the CoP's velocity) on two axes over
    N steps with the biped's kinds of costs and boxes -- input effort, velocity tracking, CoP
    centring; the CoP inside the foot at every step, the DCM inside it at the last -- ``2 N``
    unknowns, ``4 N + 4`` inequality rows.  Built on the nominal ``(A, B)``; an assembler compiled
    with ``ltv=["LIP"]`` takes per-step, per-instance ``(A_k, B_k)`` (:func:`ltv_lipm_steps`)."""
    t = api.tools
    axes = ["_x", "_y"]
    pendulum = api.ControlSystem.from_name("dP->CCC", axes, tau=tau, omega=omega)
    lip = api.ExtendedSystem.from_cotrol_system(pendulum, "x", N)
    lip.define_output("b", {"CoM": 1, "CoM_ddot": -1 / omega**2})
    lip.define_output("DCM", {"CoM": 1, "CoM_dot": 1 / omega})
    foot = t.make_simetric_vertices(np.array(foot_corner))
    form = api.Formulation()
    form.incorporate_dynamics("LIP", lip)
    form.incorporate_goal("effort", api.Cost("cCoP_dot", 1e-3, aim=[0, 0], axes=axes))
    form.incorporate_goal("velocity", api.Cost("CoM_dot", 1e-2, aim=list(target_vel), axes=axes))
    form.incorporate_goal("centre CoP", api.Cost("b", 1.0, aim=[0, 0], axes=axes))
    form.incorporate_box("CoP", api.Box.task_space("b", foot, axes))
    form.incorporate_box("terminal", api.Box.task_space("DCM", foot, axes, schedule=range(N - 1, N)))
    form.identify_qp_domain(["cCoP_dot_x", "cCoP_dot_y"])
    form.make_preview_matrices()
    return form



def ltv_lipm_steps(api, N=100, tau=0.1, omega=3.3445, theta=0.0):
    """``(A_k, B_k)`` of the ``dP->CCC`` pendulum with a slowly varying natural
    frequency ``omega_k = omega (1 + 0.05 sin(2 pi k / N + theta))``."""
    get_A, get_B, _ = api.tools.get_system_matrices("dP->CCC")
    k = np.arange(N)
    omegas = omega * (1 + 0.05 * np.sin(2 * np.pi * k / N + theta))
    A = np.stack([np.asarray(get_A(tau=tau, omega=float(o)), dtype=float) for o in omegas])
    B = np.stack([np.asarray(get_B(tau=tau, omega=float(o)), dtype=float).reshape(3, 1)
                  for o in omegas])
    return A, B
