"""Python surface of the materials the hot path supports, with the attribute names of the
reference's pybind11 classes (src/mimi/py/py_material.cpp:28-78, py_hardening.cpp:25-82):

    mat = mimi_amd.CompressibleOgdenNeoHookean(); mat.density = 1; mat.set_young_poisson(E, nu)
    mat = mimi_amd.J2(); mat.hardening = mimi_amd.JohnsonCookTemperatureAndRateDependentHardening(); ...
"""
from . import _capi


class HardeningBase:
    _kind = -1

    def name(self):
        return type(self).__name__

    def is_rate_dependent(self):
        return False


class PowerLawHardening(HardeningBase):
    _kind = 0
    sigma_y = 0.0
    n = 1.0
    eps0 = 1.0


class VoceHardening(HardeningBase):
    _kind = 1
    sigma_y = 0.0
    sigma_sat = 0.0
    strain_constant = 1.0


class JohnsonCookHardening(HardeningBase):
    _kind = 2
    A = 0.0
    B = 0.0
    n = 1.0

    def sigma_y(self):
        return self.A


class JohnsonCookRateDependentHardening(JohnsonCookHardening):
    _kind = 3
    # the reference leaves C_ uninitialised unless set (material_hardening.hpp:156); its
    # golden data correspond to 0
    C = 0.0
    eps0_dot = 1.0

    def is_rate_dependent(self):
        return True


class JohnsonCookTemperatureAndRateDependentHardening(JohnsonCookRateDependentHardening):
    _kind = 4
    reference_temperature = 0.0
    m = 1.0


class JohnsonCookConstantTemperatureHardening(JohnsonCookTemperatureAndRateDependentHardening):
    _kind = 5


class Material:
    _kind = -1
    density = -1.0
    viscosity = -1.0
    lambda_ = mu = young = poisson = K = G = -1.0

    def name(self):
        return type(self).__name__

    def set_young_poisson(self, young, poisson):
        """MaterialBase::SetYoungPoisson (materials/materials.cpp:7-14)."""
        self.young, self.poisson = float(young), float(poisson)
        self.lambda_ = young * poisson / ((1 + poisson) * (1 - 2 * poisson))
        self.mu = young / (2.0 * (1.0 + poisson))
        self.G = self.mu
        self.K = young / (3.0 * (1.0 - (2.0 * poisson)))

    def set_lame(self, lam, mu):
        """MaterialBase::SetLame (materials/materials.cpp:16-23)."""
        self.young = mu * (3 * lam + 2 * mu) / (lam + mu)
        self.poisson = lam / (2 * (lam + mu))
        self.lambda_, self.mu, self.G = float(lam), float(mu), float(mu)
        self.K = lam + 2 * mu / 3

    def _c_struct(self):
        m = _capi.Material()
        m.kind = self._kind
        m.hardening = -1
        m.density = self.density
        m.lambda_, m.mu, m.K, m.G = self.lambda_, self.mu, self.K, self.G
        return m


class CompressibleOgdenNeoHookean(Material):
    _kind = 0


class J2(Material):
    _kind = 1
    hardening = None
    heat_fraction = 0.9
    specific_heat = 0.0
    initial_temperature = 20.0
    melting_temperature = -1.0

    def _c_struct(self):
        m = super()._c_struct()
        if self.hardening is None:
            raise RuntimeError("hardening missing for " + self.name())   # materials.cpp:139-148
        h = self.hardening
        m.hardening = h._kind
        m.heat_fraction, m.specific_heat = self.heat_fraction, self.specific_heat
        m.initial_temperature, m.melting_temperature = self.initial_temperature, self.melting_temperature
        for f in ("sigma_y", "n", "eps0", "sigma_sat", "strain_constant", "A", "B", "C", "eps0_dot",
                  "reference_temperature", "m"):
            v = getattr(h, f, 0.0)
            if callable(v):
                continue
            setattr(m, f, float(v))
        return m


class StVenantKirchhoff(Material):
    """materials/materials.hpp:88-111"""
    _kind = 2


class J2Linear(Material):
    """materials/materials.hpp:142-249 (attribute names of py/py_material.cpp:47-53)"""
    _kind = 3
    isotropic_hardening = 0.0
    kinematic_hardening = 0.0
    sigma_y = 0.0

    def _c_struct(self):
        m = super()._c_struct()
        m.sigma_y = float(self.sigma_y)
        m.lin_isotropic_hardening = float(self.isotropic_hardening)
        m.lin_kinematic_hardening = float(self.kinematic_hardening)
        return m


class J2Simo(J2):
    """materials/materials.hpp:406-557"""
    _kind = 4


class J2Log(J2):
    """materials/materials.hpp:559-753"""
    _kind = 5
