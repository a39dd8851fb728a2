"""MISO settings files (the role of misopy/settings.py): an INI file whose section names do not matter
-- every option lands in one flat table -- read with literal evaluation of the values, except the
`[cluster]` section whose values stay strings (settings.py:15-59).  Defaults are the values of the
reference's shipped misopy/settings/miso_settings.txt."""
import ast
import configparser
import os

DEFAULT_SETTINGS = {"filter_results": True, "min_event_reads": 20, "cluster_command": "qsub",
                    "burn_in": 500, "lag": 10, "num_iters": 5000, "num_chains": 6,
                    "num_processors": 4}
STRAND_RULES = ("fr-unstranded", "fr-firststrand", "fr-secondstrand")


def tryEval(text):
    """The literal a setting's text spells, or the text itself."""
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


class Settings(object):
    """Process-wide settings table with the getters the sampler's callers use."""
    global_settings = dict(DEFAULT_SETTINGS)
    settings_path = None

    @classmethod
    def load(cls, path):
        cls.settings_path = path
        if path is None:
            cls.global_settings = dict(DEFAULT_SETTINGS)
            return
        if not os.path.isfile(path):
            raise IOError("Error: Settings file %s does not exist." % path)
        parser = configparser.ConfigParser()
        parser.read(path)
        cls.global_settings = {
            option: (str(value) if section == "cluster" else tryEval(value))
            for section in parser.sections() for option, value in parser.items(section)}

    @classmethod
    def get(cls):
        return cls.global_settings

    @classmethod
    def get_sampler_params(cls):
        """burn_in, lag, num_iters (required) and num_chains (6 unless set), settings.py:62-82."""
        table = cls.global_settings
        missing = [k for k in ("burn_in", "lag", "num_iters") if k not in table]
        if missing:
            raise Exception("Error: need %s parameter to be set in settings file." % missing[0])
        return {"num_chains": table.get("num_chains", 6), "burn_in": table["burn_in"],
                "lag": table["lag"], "num_iters": table["num_iters"]}

    @classmethod
    def get_min_event_reads(cls, default_min_reads=20):
        return cls.global_settings.get("min_event_reads", default_min_reads)

    @classmethod
    def get_strand_param(cls, default_strand_param="fr-unstranded"):
        rule = cls.global_settings.get("strand", default_strand_param)
        if rule not in STRAND_RULES:
            raise ValueError("Error: Invalid strand parameter %s" % rule)
        return rule

    @classmethod
    def get_num_processors(cls, default_num_processors=4):
        return int(cls.global_settings.get("num_processors", default_num_processors))


def load_settings(settings_filename):
    Settings.load(settings_filename)
    return Settings.get()
