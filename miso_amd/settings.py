"""misopy/settings.py:13-190 for Python 3: the ConfigParser settings file (section headers are
ignored, every option is global) and the getters the sampler's callers use."""
import ast
import configparser
import os

# misopy/settings/miso_settings.txt (the defaults shipped with the reference)
DEFAULT_SETTINGS = {"filter_results": True, "min_event_reads": 20, "cluster_command": "qsub",
                    "burn_in": 500, "lag": 10, "num_iters": 5000, "num_chains": 6,
                    "num_processors": 4}


def tryEval(s):
    """parse_csv.py tryEval: literal if it evaluates, else the string."""
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


class Settings(object):
    global_settings = dict(DEFAULT_SETTINGS)
    settings_path = None

    @classmethod
    def load(cls, path):
        """settings.py:15-59; path None -> the reference's shipped defaults."""
        cls.global_settings = {}
        cls.settings_path = path
        if path is None:
            cls.global_settings = dict(DEFAULT_SETTINGS)
            return
        if not os.path.isfile(path):
            raise IOError("Error: Settings file %s does not exist." % path)
        config = configparser.ConfigParser()
        config.read(path)
        for section in config.sections():
            for option in config.options(section):
                if section == "cluster":
                    cls.global_settings[option] = str(config.get(section, option))
                else:
                    cls.global_settings[option] = tryEval(config.get(section, option))

    @classmethod
    def get(cls):
        return cls.global_settings

    @classmethod
    def get_sampler_params(cls):
        """settings.py:62-82."""
        sampler_params = {'num_chains': 6}
        for name in ['burn_in', 'lag', 'num_iters']:
            if name not in cls.global_settings:
                raise Exception("Error: need %s parameter to be set in settings file." % name)
            sampler_params[name] = cls.global_settings[name]
        if 'num_chains' in cls.global_settings:
            sampler_params['num_chains'] = cls.global_settings['num_chains']
        return sampler_params

    @classmethod
    def get_min_event_reads(cls, default_min_reads=20):
        return cls.global_settings.get("min_event_reads", default_min_reads)

    @classmethod
    def get_strand_param(cls, default_strand_param="fr-unstranded"):
        """settings.py:129-144."""
        strandedness = cls.global_settings.get("strand", default_strand_param)
        if strandedness not in ("fr-unstranded", "fr-firststrand", "fr-secondstrand"):
            raise ValueError("Error: Invalid strand parameter %s" % strandedness)
        return strandedness

    @classmethod
    def get_num_processors(cls, default_num_processors=4):
        return int(cls.global_settings.get("num_processors", default_num_processors))


def load_settings(settings_filename):
    Settings.load(settings_filename)
    return Settings.get()
