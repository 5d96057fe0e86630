"""Host side of MOM_sum_output (src/diagnostics/MOM_sum_output.F90): the bookkeeping and the text of `ocean.stats`.

The reference's regression suite defines the model state by this file (.testing/README.rst:293; the test.* targets
`diff` two ocean.stats files, .testing/Makefile:449-454).  The globally summed numbers come from the device
(Dycore.write_energy -> mom6x_write_energy); what write_energy does around them lives here, as it lives on the host in
the reference: the values of the previous call (mass_prev_EFP :565, salt / heat :735), the differences in extended fixed
point (EFP_minus, MOM_coms.F90:772), the derived ratios (:769-777) and the formatted lines (:616-632 header, :871-905).
The write schedule (ENERGYSAVEDAYS :472-497) is the caller's: every call of `record` writes a line.
"""
import math

_PREC = 1 << 46                                  # MOM_coms.F90:30
_PR = [2.0 ** 92, 2.0 ** 46, 1.0, 2.0 ** -46, 2.0 ** -92, 2.0 ** -138]     # :40-41


def _trunc_div(a, b):
    q = abs(a) // b
    return q if a >= 0 else -q


def EFP_increment(a, b):
    """increment_ints (MOM_coms.F90:618-648): a + b, limb by limb, with the one-step carry of the reference."""
    s = [int(x) for x in a]
    for i in range(5, 0, -1):
        s[i] += int(b[i])
        if s[i] > _PREC:
            s[i] -= _PREC; s[i - 1] += 1
        elif s[i] < -_PREC:
            s[i] += _PREC; s[i - 1] -= 1
    s[0] += int(b[0])
    return s


def EFP_minus(a, b):
    """EFP_minus (:772-782)."""
    return EFP_increment([-int(x) for x in b], a)


def EFP_to_real(a):
    """EFP_to_real (:797-803): regularize_ints (:709-747), then ints_to_real (:605-614)."""
    s = [int(x) for x in a]
    for i in range(5, 0, -1):
        if abs(s[i]) >= _PREC:
            c = _trunc_div(s[i], _PREC)
            s[i] -= c * _PREC; s[i - 1] += c
    positive = True
    for x in s:
        if x != 0:
            positive = x > 0
            break
    for i in range(5, 0, -1):
        if positive and s[i] < 0:
            s[i] += _PREC; s[i - 1] -= 1
        elif (not positive) and s[i] > 0:
            s[i] -= _PREC; s[i - 1] += 1
    r = 0.0
    for i in range(6):
        r = r + _PR[i] * float(s[i])
    return r


def fortran_es(x, w, d):
    """Fortran ESw.d."""
    if math.isnan(x):
        return "NaN".rjust(w)
    if math.isinf(x):
        return ("-Infinity" if x < 0 else "Infinity").rjust(w) if w >= 9 else "*" * w
    m, e = ("%.*E" % (d, x)).split("E")
    e = int(e)
    s = "%sE%+03d" % (m, e) if abs(e) < 100 else "%s%+04d" % (m, e)
    return s.rjust(w) if len(s) <= w else "*" * w


def fortran_f(x, w, d):
    """Fortran Fw.d."""
    s = "%.*f" % (d, x)
    if len(s) > w and s.startswith("0."):
        s = s[1:]
    elif len(s) > w and s.startswith("-0."):
        s = "-" + s[2:]
    return s.rjust(w) if len(s) <= w else "*" * w


class SumOutput:
    """Sum_output_CS of one run.  use_temperature selects the twelve-column layout; timeunit is TIMEUNIT [s]."""

    def __init__(self, use_temperature=False, timeunit=86400.0, C_p=3991.86795711963):
        self.use_temperature = bool(use_temperature)
        self.timeunit = timeunit
        self.C_p = C_p
        self.previous_calls = 0
        self.ntrunc = 0
        self.mass_prev_EFP = None; self.salt_prev_EFP = None; self.heat_prev_EFP = None
        self.fresh_water_in_EFP = [0] * 6; self.net_salt_in_EFP = [0] * 6; self.net_heat_in_EFP = [0] * 6
        self.lines = []

    def header(self):
        """The two header lines written when the file is created (:616-660)."""
        days = abs(self.timeunit - 86400.0) < 1.0
        if days:
            first = "  Step," + " " * 7 + "Day,  Truncs,      "
            t2 = " " * 12 + "[days]" + " " * 17
        else:
            first = "  Step," + " " * 7 + "Time, Truncs,      "
            if 0.99 <= self.timeunit < 1.01:
                tu = "           [seconds]     "
            elif 3599.0 <= self.timeunit < 3601.0:
                tu = "            [hours]      "
            elif 86399.0 <= self.timeunit < 86401.0:
                tu = "             [days]      "
            elif 3.0e7 <= self.timeunit < 3.2e7:
                tu = "            [years]      "
            else:
                tu = " " * 9 + "[" + fortran_es(self.timeunit, 8, 2) + " s]    "
            t2 = tu[:25].rjust(25) + " " * 10
        if self.use_temperature:
            l1 = first + "Energy/Mass,      Maximum CFL,  Mean Sea Level,  Total Mass,  Mean Salin, Mean Temp, Frac Mass Err,   Salin Err,    Temp Err"
            l2 = (t2 + "[m2 s-2]" + " " * 11 + "[Nondim]" + " " * 7 + "[m]" + " " * 13 + "[kg]" + " " * 9 + "[PSU]" + " " * 6 + "[degC]" +
                  " " * 7 + "[Nondim]" + " " * 8 + "[PSU]" + " " * (8 if days else 6) + "[degC]")
        else:
            l1 = first + "Energy/Mass,      Maximum CFL,  Mean sea level,   Total Mass,    Frac Mass Err"
            l2 = t2 + "[m2 s-2]" + " " * 11 + "[Nondim]" + " " * 8 + "[m]" + " " * 13 + "[kg]" + " " * 11 + "[Nondim]"
        return [l1, l2]

    def record(self, sums, day_seconds, n):
        """write_energy from :565 on, given the sums (Dycore.write_energy) at model time day_seconds [s] and step n.
        Returns (stdout line, ocean.stats line); the latter is also appended to self.lines (after the header)."""
        mass_EFP = [int(x) for x in sums["mass_EFP"]]
        if self.previous_calls == 0:
            self.mass_prev_EFP = mass_EFP
            self.fresh_water_in_EFP = [0] * 6
            if self.use_temperature:
                self.net_salt_in_EFP = [0] * 6; self.net_heat_in_EFP = [0] * 6
            if day_seconds <= 0.0:
                self.lines.extend(self.header())
        mass_tot = sums["mass_tot"]
        Salt = Heat = salin = salin_anom = temp = temp_anom = 0.0
        if self.use_temperature:
            salt_EFP = [int(x) for x in sums["salt_EFP"]]; heat_EFP = [int(x) for x in sums["heat_EFP"]]
            Salt = EFP_to_real(salt_EFP); Heat = EFP_to_real(heat_EFP)                       # :730-731 (kg_to_RZL2 = 1)
            if self.previous_calls == 0:
                self.salt_prev_EFP = salt_EFP; self.heat_prev_EFP = heat_EFP
            Salt_chg_EFP = EFP_minus(salt_EFP, self.salt_prev_EFP)
            Salt_anom = EFP_to_real(EFP_minus(Salt_chg_EFP, self.net_salt_in_EFP))
            Heat_chg_EFP = EFP_minus(heat_EFP, self.heat_prev_EFP)
            Heat_anom = EFP_to_real(EFP_minus(Heat_chg_EFP, self.net_heat_in_EFP))
        mass_chg_EFP = EFP_minus(mass_EFP, self.mass_prev_EFP)                               # :749-755
        mass_anom = EFP_to_real(EFP_minus(mass_chg_EFP, self.fresh_water_in_EFP))
        if self.use_temperature:
            salin = Salt / mass_tot; salin_anom = Salt_anom / mass_tot                      # :757-762
            temp = Heat / (mass_tot * self.C_p); temp_anom = Heat_anom / (mass_tot * self.C_p)
        toten = sums["KE_tot"] + sums["PE_tot"]
        En_mass = toten / mass_tot
        start_of_day = int(day_seconds) % 86400; num_days = int(day_seconds) // 86400
        if abs(self.timeunit - 86400.0) < 1.0:
            reday = float(num_days) + (float(start_of_day) / 86400.0); intro = "MOM Day"
        else:
            reday = float(num_days) * (86400.0 / self.timeunit) + float(start_of_day) / abs(self.timeunit); intro = "MOM Time"
        if reday < 1.0e8:
            day_str = fortran_f(reday, 12, 3)
        elif reday < 1.0e11:
            day_str = fortran_f(reday, 15, 3)
        else:
            day_str = fortran_es(reday, 15, 9)
        n_str = ("%6d" if n < 1000000 else "%7d" if n < 10000000 else "%8d" if n < 100000000 else "%10d") % n
        date_str = intro + day_str.rstrip()
        max_CFL = sums["max_CFL"]
        out = date_str + " " + n_str.rstrip() + ": En " + fortran_es(En_mass, 12, 6) + ", MaxCFL " + fortran_f(max_CFL[0], 8, 5) + \
            ", Mass " + fortran_es(mass_tot, 18, 12)
        line = n_str.rstrip() + "," + day_str.rstrip() + "," + "%6d" % self.ntrunc + ", En " + fortran_es(En_mass, 22, 16) + \
            ", CFL " + fortran_f(max_CFL[0], 8, 5) + ", SL " + fortran_es(-sums["Z_0APE"][0], 11, 4)
        if self.use_temperature:
            out += ", Salt " + fortran_f(salin, 15, 11) + ", Temp " + fortran_f(temp, 15, 11)
            line += ", M " + fortran_es(mass_tot, 11, 5) + ", S" + fortran_f(salin, 8, 4) + ", T" + fortran_f(temp, 8, 4) + \
                ", Me " + fortran_es(mass_anom / mass_tot, 9, 2) + ", Se " + fortran_es(salin_anom, 9, 2) + ", Te " + fortran_es(temp_anom, 9, 2)
        else:
            line += ", Mass " + fortran_es(mass_tot, 11, 5) + ", Me " + fortran_es(mass_anom / mass_tot, 9, 2)
        self.lines.append(line)
        if math.isnan(En_mass):
            raise RuntimeError("write_energy : NaNs in total model energy forced model termination.")
        # :1013-1018: this call's totals become the previous ones
        self.ntrunc = 0
        self.previous_calls += 1
        self.mass_prev_EFP = mass_EFP; self.fresh_water_in_EFP = [0] * 6
        if self.use_temperature:
            self.salt_prev_EFP = salt_EFP; self.heat_prev_EFP = heat_EFP
            self.net_salt_in_EFP = [0] * 6; self.net_heat_in_EFP = [0] * 6
        return out, line

    def write(self, path):
        with open(path, "w") as f:
            f.write("\n".join(self.lines) + "\n")
