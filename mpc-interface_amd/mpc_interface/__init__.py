"""Drop-in ``mpc_interface`` package: the reference's problem-description API
(``Formulation``, ``Cost``, ``Constraint``, ``Box``, ``ExtendedSystem``,
``DomainVariable``, ``ControlSystem``, ``LineCombo``, ``tools``) with the QP
assembly executed by hand-written HIP kernels for MI355X (see ``mpcasm``)."""
