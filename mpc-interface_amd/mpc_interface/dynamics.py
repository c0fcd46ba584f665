"""Dynamics description: DomainVariable, ExtendedSystem, ControlSystem.

Host-side mirror of /root/reference/python/mpc_interface/dynamics.py (same
constructor signatures, attribute names and update semantics).  These objects
hold *structure* (names -> IDs, sizes, definitions) and the horizon matrices
``matrices = U + [S]``; the numeric production of ``S, U`` from ``(A, B)`` is
the HIP kernel behind :func:`mpc_interface.tools.extend_matrices`.
"""
from collections.abc import Iterable

import numpy as np

from . import tools as use
from .combinations import LineCombo


def _axes_list(axes):
    """Normalise the ``axes`` argument (dynamics.py:49-56, :187-197)."""
    if axes is None:
        return [""]
    if isinstance(axes, str):
        return [axes]
    if not isinstance(axes, Iterable):
        return [str(axes)]
    return axes


def _names_list(names, message):
    if isinstance(names, str):
        names = [names]
    for name in names:
        if not isinstance(name, str):
            raise TypeError(message)
    return names


class DomainVariable:
    """Free variables of the QP domain (dynamics.py:29-123).

    ``names``/``sizes`` give one entry per variable; each is replicated over
    ``axes``.  Every variable is defined as the identity of itself.
    """

    def __init__(self, names, sizes, axes=None, time_variant=False,
                 how_to_update_size=None):
        names = _names_list(names, "all variable names must be strings.")
        if not isinstance(sizes, Iterable):
            sizes = [sizes]
        if len(names) != len(sizes):
            raise IndexError(
                "'names' and 'sizes' must have the same amount of elements."
            )
        self.axes = _axes_list(axes)

        self.identify_domain(names)
        self.set_sizes(names, sizes)

        self.outputs = []
        self.definitions = {}
        self.make_definitions()

        self.time_variant = time_variant
        self._resize = (
            how_to_update_size
            if (how_to_update_size is not None and time_variant)
            else use.do_not_update
        )

    def identify_domain(self, names):
        self.domain_ID = {
            name + axis: k for axis in self.axes for k, name in enumerate(names)
        }

    def set_sizes(self, names, sizes):
        self.domain = {var: sizes[k] for var, k in self.domain_ID.items()}
        self.all_variables = self.domain

    def update_sizes(self, **kargs):
        self._resize(self, **kargs)

    def make_definitions(self):
        for var, size in self.domain.items():
            combo = LineCombo({var: np.eye(size)})
            combo._coefficients = ["I"]
            self.definitions[var] = combo

    def define_output(self, name, combination, time_variant=False,
                      how_to_update=None):
        """Extra per-axis definitions (dynamics.py:92-112); the update callback
        receives ``domVar=self``."""
        for axis in self.axes:
            per_axis = {var + axis: coef for var, coef in combination.items()}
            self.definitions[name + axis] = LineCombo(
                per_axis, time_variant=time_variant, how_to_update=how_to_update
            )
            self.outputs.append(name + axis)

    def update_definitions(self):
        for var, size in self.domain.items():
            self.definitions[var].matrices[0] = np.eye(size)
        for output in self.outputs:
            self.definitions[output].update(domVar=self)

    def update(self, **kargs):
        self.update_sizes(**kargs)
        self.update_definitions()


class ExtendedSystem:
    """Horizon-extended dynamics ``x = S x0 + sum_j U_j u_j`` (dynamics.py:131-344).

    ``S`` has shape ``[N, n, n]`` and ``U`` is a list of ``m`` arrays
    ``[N, p_u, n]``; a bare array is accepted for one input, and 2-D arrays
    when there is a single state.  ``matrices = U + [S]`` (the order matters:
    ``domain_ID`` indexes into it).
    """

    def __init__(self, input_names, state_names, state_vector_name, S, U,
                 axes=None, time_variant=False, how_to_update_matrices=None):
        input_names = _names_list(input_names, "all input names must be strings.")
        state_names = _names_list(state_names, "all variable names must be strings.")
        if not isinstance(state_vector_name, str):
            raise TypeError("the 'state_vector_name' must be a single string.")

        if not isinstance(U, list) and len(input_names) == 1:
            U = [U]
        if len(state_names) == 1:
            if S.ndim == 2:
                S = S[:, :, None]
            U = [u[:, :, None] if u.ndim == 2 else u for u in U]

        self.axes = _axes_list(axes)
        self.matrices = list(U) + [S]
        self.state_vector_name = state_vector_name

        self.identify_domain(input_names, state_names)
        self.set_sizes()

        self.outputs = []
        self.definitions = {}
        self.make_definitions()

        self.time_variant = time_variant
        self._refresh = (
            how_to_update_matrices
            if (how_to_update_matrices is not None and time_variant)
            else use.do_not_update
        )

    @classmethod
    def from_cotrol_system(cls, control_system, state_vector_name, horizon_lenght):
        """Extend a :class:`ControlSystem` over the horizon (dynamics.py:216-246).

        The name keeps the reference's spelling; for time-variant control
        systems the extension is redone at every ``update`` (keyword
        ``control_system`` must then be passed, dynamics.py:222-231).
        """
        S, U = use.extend_matrices(horizon_lenght, control_system.A, control_system.B)

        refresh = None
        if control_system.time_variant:

            def refresh(ext_syst, **kargs):
                source = kargs["control_system"]
                source.update_matrices(**kargs)
                S_new, U_new = use.extend_matrices(horizon_lenght, source.A, source.B)
                ext_syst.matrices = U_new + [S_new]

        return cls(
            control_system.input_names,
            control_system.state_names,
            state_vector_name,
            S,
            U,
            control_system.axes,
            control_system.time_variant,
            refresh,
        )

    from_control_system = from_cotrol_system

    def identify_domain(self, input_names, state_names):
        """names -> index into ``matrices`` / into the state axis (dynamics.py:248-258)."""
        slots = {name: k for k, name in enumerate(input_names)}
        slots[self.state_vector_name + "0"] = len(input_names)
        self.domain_ID = {
            name + axis: k for axis in self.axes for name, k in slots.items()
        }
        self.state_ID = {
            name + axis: k for axis in self.axes for k, name in enumerate(state_names)
        }

    def set_sizes(self):
        horizon = self.matrices[-1].shape[0]
        self.domain = {
            var: self.matrices[k].shape[1] for var, k in self.domain_ID.items()
        }
        self.all_variables = dict(self.domain)
        self.all_variables.update({state: horizon for state in self.state_ID})

    def update_sizes(self):
        if self.time_variant:
            for var, k in self.domain_ID.items():
                self.domain[var] = self.matrices[k].shape[1]
            self.all_variables.update(self.domain)

    def _state_terms(self, state):
        """Domain variables feeding a state (axis matched on the last two
        characters, dynamics.py:284-294)."""
        if self.axes == [""]:
            return list(self.domain_ID.items())
        return [(v, k) for v, k in self.domain_ID.items() if v[-2:] == state[-2:]]

    def make_definitions(self):
        for var, size in self.domain.items():
            combo = LineCombo({var: np.eye(size)})
            combo._coefficients = ["I"]
            self.definitions[var] = combo

        n_inputs = len(self.matrices) - 1
        for state, sID in self.state_ID.items():
            combo = LineCombo(
                {v: self.matrices[k][..., sID] for v, k in self._state_terms(state)}
            )
            combo._coefficients = ["U"] * n_inputs + ["S"]
            self.definitions[state] = combo

    def define_output(self, name, combination, time_variant=False,
                      how_to_update=None):
        """Extra per-axis definitions (dynamics.py:300-322); the update callback
        receives ``extSyst=self``."""
        for axis in self.axes:
            per_axis = {var + axis: coef for var, coef in combination.items()}
            combo = LineCombo(
                per_axis, time_variant=time_variant, how_to_update=how_to_update
            )
            self.definitions[name + axis] = combo
            if time_variant:
                combo.update(extSyst=self)
            self.outputs.append(name + axis)

    def update_definitions(self):
        for var, size in self.domain.items():
            self.definitions[var].matrices[0] = np.eye(size)

        for state, sID in self.state_ID.items():
            combo = self.definitions[state]
            for pos, var in enumerate(combo.variables):
                combo.matrices[pos] = self.matrices[self.domain_ID[var]][..., sID]

        for output in self.outputs:
            self.definitions[output].update(extSyst=self)

    def update_matrices(self, **kargs):
        self._refresh(self, **kargs)

    def update(self, **kargs):
        if self.time_variant:
            self.update_matrices(**kargs)
            self.update_sizes()
            self.update_definitions()


class ControlSystem:
    """One-step model ``x+ = A x + B u`` (dynamics.py:347-407)."""

    def __init__(self, input_names, state_names, A, B, axes=None,
                 time_variant=False, how_to_update_matrices=None):
        self.state_names = state_names
        self.input_names = input_names
        self.A = A
        self.B = B
        self.axes = axes
        self.time_variant = time_variant
        self._refresh = (
            how_to_update_matrices
            if (how_to_update_matrices is not None and time_variant)
            else use.do_not_update
        )

    def update_matrices(self, **kargs):
        self._refresh(self, **kargs)

    @staticmethod
    def check_system_parameters(system_name):
        return use.get_system_matrices(system_name)[2]

    @classmethod
    def from_name(cls, system_name, axes=None, time_variant=False,
                  how_to_update_matrices=None, **kargs):
        """Named LIPM variants (dynamics.py:379-407)."""
        input_names, state_names = use.get_system_variables(system_name)
        get_A, get_B, parameters = use.get_system_matrices(system_name)
        try:
            A = get_A(**kargs)
            B = get_B(**kargs)
        except Exception:
            raise ValueError("This system formulation requires: " + str(parameters))

        system = cls(input_names, state_names, A, B, axes, time_variant,
                     how_to_update_matrices)
        system.get_A = get_A
        system.get_B = get_B
        system.parameters = kargs
        system.system_name = system_name
        return system
