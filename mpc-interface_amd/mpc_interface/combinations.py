"""Linear-combination container of the problem description.

Host-side mirror of the reference's ``LineCombo``
(/root/reference/python/mpc_interface/combinations.py:4-48): an ordered list of
``(variable name, coefficient)`` pairs plus an optional update callback.  It is
pure structure: the plan compiler (``mpcasm.plan``) reads ``variables`` and
``matrices`` to flatten the definition graph into device tables.
"""
from . import tools as _tools


class LineCombo:
    """``sum_k matrices[k] * variables[k]`` (reference combinations.py:5-22).

    ``combination`` is a dict ``{variable: coefficient}``; a coefficient is a
    scalar, a 1-D array or a 2-D array.  ``how_to_update(combo, **kargs)`` is
    only honoured when ``time_variant`` is true.
    """

    def __init__(self, combination=None, data=None, how_to_update=None,
                 time_variant=False):
        self.variables = []
        self.matrices = []
        if combination is not None:
            for name, coefficient in combination.items():
                self.variables.append(name)
                self.matrices.append(coefficient)

        self.data = data if data is not None else []
        self.time_variant = time_variant
        self._updater = (
            how_to_update
            if (how_to_update is not None and time_variant)
            else _tools.do_not_update
        )
        self._coefficients = ["C%d" % k for k in range(len(self.variables))]

    # reference combinations.py:24-25
    def update(self, **kargs):
        self._updater(self, **kargs)

    # mapping-like access, reference combinations.py:27-37
    def __getitem__(self, variable):
        return self.matrices[self.variables.index(variable)]

    def items(self):
        return zip(self.variables, self.matrices)

    def keys(self):
        return self.variables

    def values(self):
        return self.matrices

    def __repr__(self):
        return " + ".join(
            "%s ( %s )" % (c, v) for c, v in zip(self._coefficients, self.variables)
        )

    __str__ = __repr__
